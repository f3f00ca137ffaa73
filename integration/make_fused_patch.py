"""Regenerates integration/fused_callsites.patch from the reference checkout (maintainer tooling).

    python integration/make_fused_patch.py [/root/reference]

The patch is the MINIMAL change that routes `SplatfactoModel.get_outputs` / `render_gaussian_attrs`
(street_gaussians_ns/sgn_splatfacto.py:857-873, 933-996) onto `sgn_rast.fused`: activations (exp, quaternion
normalisation, sigmoid), the SH concatenation, the view directions and the depth pass move into the kernels; nothing
else of the file changes.  The scene graph's sub-model passes keep calling `render_gaussian_attrs` with concatenated
colours and take the original SH branch (an `isinstance` test), but gain the fused rasterization.
"""
import difflib
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
REL = "street_gaussians_ns/sgn_splatfacto.py"
HERE = os.path.dirname(os.path.abspath(__file__))

EDITS = [
    # imports
    ("from gsplat.sh import num_sh_bases, spherical_harmonics\n",
     "from gsplat.sh import num_sh_bases, spherical_harmonics\n"
     "from sgn_rast import fused as sgn_fused  # MI355X: activations / concat / depth pass folded into the kernels\n"),
    # get_outputs: raw parameters go to the fused projection; no concatenated SH copy
    ("        scales_crop = torch.exp(scales_crop)\n"
     "        colors_crop = torch.cat((features_dc_crop, features_rest_crop), dim=1)\n",
     "        colors_crop = (features_dc_crop, features_rest_crop)  # sgn_fused: un-concatenated SH leaves\n"),
    ("self.num_tiles_hit, _ = project_gaussians(  # type: ignore\n"
     "            means_crop,\n"
     "            scales_crop,\n"
     "            1,\n"
     "            quats_crop / quats_crop.norm(dim=-1, keepdim=True),\n",
     "self.num_tiles_hit, _ = sgn_fused.project_gaussians_fused(  # exp / normalise in-kernel\n"
     "            means_crop,\n"
     "            scales_crop,\n"
     "            quats_crop,\n"),
    # render_gaussian_attrs: fused SH for the un-concatenated leaves
    ("        if self.config.sh_degree > 0:\n"
     "            viewdirs = means.detach() - camera.camera_to_worlds.detach()[..., :3, 3]  # (N, 3)\n",
     "        if self.config.sh_degree > 0 and isinstance(colors, tuple):\n"
     "            n = min(self.step // self.config.sh_degree_interval, self.config.sh_degree)\n"
     "            if not self.training:\n"
     "                n = self.config.sh_degree\n"
     "            rgbs = sgn_fused.spherical_harmonics_fused(  # view dirs, concat, SH, +0.5, clamp in one pass\n"
     "                n, means, camera.camera_to_worlds.detach()[0, :3, 3], colors[0], colors[1])\n"
     "        elif self.config.sh_degree > 0:\n"
     "            viewdirs = means.detach() - camera.camera_to_worlds.detach()[..., :3, 3]  # (N, 3)\n"),
    ("            rgbs = torch.sigmoid(colors[:, 0, :])\n",
     "            rgbs = torch.sigmoid((colors[0] if isinstance(colors, tuple) else colors)[:, 0, :])\n"),
    # opacities stay logits: the sigmoid runs inside the rasterizer's row build / gradient unpack
    ("            opacities = torch.sigmoid(opacities) #* comp[:, None]\n",
     "            pass  # sgn_fused: sigmoid in-kernel\n"),
    ("            opacities = torch.sigmoid(opacities)\n",
     "            pass  # sgn_fused: sigmoid in-kernel\n"),
    ("            rgb, alpha = rasterize_gaussians(  # type: ignore\n",
     "            rgb, alpha, depth_channel = sgn_fused.rasterize_gaussians_fused(  # depth rides as a 4th channel\n"),
    ("                background=background,\n"
     "                return_alpha=True,\n"
     "            )  # type: ignore\n",
     "                background=background,\n"
     "                return_alpha=True,\n"
     "                depth_channel=True,\n"
     "            )  # type: ignore\n"),
    ("            depth_im = rasterize_gaussians(\n"
     "                xys,\n"
     "                depths,\n"
     "                radii,\n"
     "                conics,\n"
     "                num_tiles_hit,\n"
     "                depths[:, None].repeat(1, 3),\n"
     "                opacities,\n"
     "                H,\n"
     "                W,\n"
     "                self.config.block_width,\n"
     "                torch.zeros(3, device=self.device),\n"
     "            )[..., 0:1]\n",
     "            depth_im = depth_channel[..., None]  # accumulated by the colour pass (no second rasterization)\n"),
]


def patched(src: str) -> str:
    for old, new in EDITS:
        assert src.count(old) == 1, ("edit does not apply exactly once", old[:60], src.count(old))
        src = src.replace(old, new)
    return src


def main():
    src = open(os.path.join(REF, REL)).read()
    new = patched(src)
    diff = "".join(difflib.unified_diff(src.splitlines(True), new.splitlines(True), "a/" + REL, "b/" + REL, n=2))
    open(os.path.join(HERE, "fused_callsites.patch"), "w").write(diff)
    print(f"{diff.count(chr(10))} lines of patch, {len(EDITS)} edits")


if __name__ == "__main__":
    main()
