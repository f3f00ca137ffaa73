"""open3d stand-in: only names that appear in annotations / LiDAR helpers the tests never call."""
class _NS:
    def __getattr__(self, name):
        return type(name, (), {})
geometry, utility, io = _NS(), _NS(), _NS()
