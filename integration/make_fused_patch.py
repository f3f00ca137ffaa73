"""Regenerates integration/fused_callsites.patch and integration/fused_scene_graph.patch from the reference checkout
(maintainer tooling).

    python integration/make_fused_patch.py [/root/reference]

`fused_callsites.patch` is the MINIMAL change that routes `SplatfactoModel.get_outputs` / `render_gaussian_attrs`
(street_gaussians_ns/sgn_splatfacto.py:857-873, 933-996) onto `sgn_rast.fused`: activations (exp, quaternion
normalisation, sigmoid), the SH concatenation, the view directions and the depth pass move into the kernels; nothing
else of the file changes.  On its own it already serves the scene graph: its sub-model passes keep calling
`render_gaussian_attrs` with concatenated colours and take the original SH branch (an `isinstance` test), but gain the
fused rasterization.

`fused_scene_graph.patch` (applied on top of it) does the same for the model the reference SHIPS
(`SplatfactoSceneGraphModel`, sgn_config.py:42): the aggregation of sgn_splatfacto_scene_graph.py:332-360 — per-object
`object2world_gs` (matmul, add, quaternion product), the Fourier DC sum — moves into the projection / SH kernels
(per-Gaussian object id + small per-step pose / weight tables), and the two sub-model accumulation passes (:364-366)
become id windows of the main pass's depth list instead of ~20 concatenations, two wasted SH evaluations and a
re-binning each.
"""
import difflib
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
REL = "street_gaussians_ns/sgn_splatfacto.py"
REL_SG = "street_gaussians_ns/sgn_splatfacto_scene_graph.py"
HERE = os.path.dirname(os.path.abspath(__file__))

EDITS = [
    # imports
    ("from gsplat.sh import num_sh_bases, spherical_harmonics\n",
     "from gsplat.sh import num_sh_bases, spherical_harmonics\n"
     "from sgn_rast import fused as sgn_fused  # MI355X: activations / concat / depth pass folded into the kernels\n"),
    # get_outputs: raw parameters go to the fused projection; no concatenated SH copy
    ("        scales_crop = torch.exp(scales_crop)\n"
     "        colors_crop = torch.cat((features_dc_crop, features_rest_crop), dim=1)\n",
     "        colors_crop = (features_dc_crop, features_rest_crop)  # sgn_fused: un-concatenated SH leaves\n"
     "        self.sgn_main_attrs, self.sgn_group_acc = None, []\n"),
    ("self.num_tiles_hit, _ = project_gaussians(  # type: ignore\n"
     "            means_crop,\n"
     "            scales_crop,\n"
     "            1,\n"
     "            quats_crop / quats_crop.norm(dim=-1, keepdim=True),\n",
     "self.num_tiles_hit, _ = sgn_fused.project_gaussians_fused(  # exp / normalise in-kernel\n"
     "            means_crop,\n"
     "            scales_crop,\n"
     "            quats_crop,\n"),
    # (scene graph, fused_scene_graph.patch: per-Gaussian object ids + pose table; absent -> no rigid transform)
    ("            self.config.block_width,\n"
     "        )  # type: ignore\n"
     "\n"
     "        if self.config.use_sky_sphere:\n"
     "            sky_capture = self.env_map(camera, self.training)\n",
     "            self.config.block_width,\n"
     "            object_ids=(getattr(self, \"sgn_tables\", None) or {}).get(\"object_ids\"),\n"
     "            poses=(getattr(self, \"sgn_tables\", None) or {}).get(\"poses\"),\n"
     "        )  # type: ignore\n"
     "\n"
     "        if self.config.use_sky_sphere:\n"
     "            sky_capture = self.env_map(camera, self.training)\n"),
    ("        output_names = ['rgb', 'accumulation', 'depth']\n",
     "        self.sgn_main_attrs = gaussian_attrs  # the scene graph's sub-model passes are id windows of this pass\n"
     "        output_names = ['rgb', 'accumulation', 'depth']\n"),
    ("        out = self.render_gaussian_attrs(camera, gaussian_attrs, output_names)\n",
     "        out = self.render_gaussian_attrs(camera, gaussian_attrs, output_names,\n"
     "                                         group_split=getattr(self, \"sgn_group_split\", None))\n"),
    ("    def render_gaussian_attrs(self, camera: Cameras, gaussian_attrs: Dict[str, Union[torch.Tensor, List]], "
     "output_names: List[str]=[]) -> Dict[str, Union[torch.Tensor, List]]:\n",
     "    def render_gaussian_attrs(self, camera: Cameras, gaussian_attrs: Dict[str, Union[torch.Tensor, List]], "
     "output_names: List[str]=[], id_range=None, group_split=None) -> Dict[str, Union[torch.Tensor, List]]:\n"),
    # render_gaussian_attrs: fused SH for the un-concatenated leaves
    ("        if self.config.sh_degree > 0:\n"
     "            viewdirs = means.detach() - camera.camera_to_worlds.detach()[..., :3, 3]  # (N, 3)\n",
     "        if self.config.sh_degree > 0 and isinstance(colors, tuple) and id_range is not None and output_names == ['accumulation']:\n"
     "            rgbs = means.new_zeros(means.shape[0], 3)  # an accumulation-only id window: colour is irrelevant\n"
     "        elif self.config.sh_degree > 0 and isinstance(colors, tuple):\n"
     "            n = min(self.step // self.config.sh_degree_interval, self.config.sh_degree)\n"
     "            if not self.training:\n"
     "                n = self.config.sh_degree\n"
     "            sgn_t = getattr(self, \"sgn_tables\", None) or {}\n"
     "            rgbs = sgn_fused.spherical_harmonics_fused(  # view dirs, Fourier DC, concat, SH, +0.5, clamp in one pass\n"
     "                n, means, camera.camera_to_worlds.detach()[0, :3, 3], colors[0], colors[1],\n"
     "                object_ids=sgn_t.get(\"object_ids\"), idft=sgn_t.get(\"idft\"), poses=sgn_t.get(\"poses\"))\n"
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
     "            # sgn_fused: depth rides as a 4th channel; with group_split so do the accumulations of ids < / >= the split\n"
     "            sgn_out = sgn_fused.rasterize_gaussians_fused(\n"),
    ("                background=background,\n"
     "                return_alpha=True,\n"
     "            )  # type: ignore\n",
     "                background=background,\n"
     "                return_alpha=True,\n"
     "                depth_channel='depth' in output_names,   # (a sub-model accumulation pass asks for no depth image)\n"
     "                id_range=id_range,\n"
     "                group_split=group_split,\n"
     "            )  # type: ignore\n"
     "            rgb, alpha = sgn_out[0], sgn_out[1]   # (img, alpha[, depth][, acc_head, acc_tail])\n"
     "            depth_channel, self.sgn_group_acc = (sgn_out[2] if len(sgn_out) > 2 else None), list(sgn_out[3:])\n"),
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


EDITS_SG = [
    ("from pytorch3d.transforms import quaternion_multiply\n",
     "from pytorch3d.transforms import quaternion_multiply\n"
     "from sgn_rast import fused as sgn_fused  # MI355X: the aggregation below folded into the projection / SH kernels\n"),
    # the Fourier DC sum moves into the SH kernel: only the frame's inverse-DFT weights are needed here
    ("    def get_fourier_features(self, frame, trackId, obj_model: SplatfactoModel):\n",
     "    def get_fourier_features(self, frame, trackId, obj_model: SplatfactoModel, weights_only=False):\n"),
    ("        idft_base = IDFT(t, obj_model.config.fourier_features_dim).to(self.device)\n",
     "        if weights_only:  # sgn_fused: sum_f features_dc[:, f] * idft[f] runs inside the SH kernel\n"
     "            return sgn_fused.memo(IDFT, float(t), obj_model.config.fourier_features_dim)[0]\n"
     "        idft_base = IDFT(t, obj_model.config.fourier_features_dim).to(self.device)\n"),
    # get_submodel_output: a sub-model pass is an id window of the main pass (same projection, same depth list)
    ("        if object_means is None:\n"
     "            submodel_means = self.aggregate_submodel_var(\"means\", submodel_names)\n"
     "            submodel_features_dc = self.aggregate_submodel_var(\"features_dc\", submodel_names)\n"
     "        else:\n",
     "        if object_means is not None:\n"),
    ("            submodel_means = torch.cat(object_means, dim=0)\n"
     "            submodel_features_dc = torch.cat(object_features_dc, dim=0)\n"
     "        submodel_opacities = self.aggregate_submodel_var(\"opacities\", submodel_names)\n"
     "        submodel_features_rest = self.aggregate_submodel_var(\"features_rest\", submodel_names)\n"
     "        submodel_xys = self.aggregate_submodel_var(\"xys\", submodel_names)\n"
     "        submodel_depths = self.aggregate_submodel_var(\"depths\", submodel_names)\n"
     "        submodel_radii = self.aggregate_submodel_var(\"radii\", submodel_names)\n"
     "        submodel_conics = self.aggregate_submodel_var(\"conics\", submodel_names)\n"
     "        submodel_num_tiles_hit = self.aggregate_submodel_var(\"num_tiles_hit\", submodel_names)\n"
     "        # render submodel\n"
     "        colors = torch.cat((submodel_features_dc, submodel_features_rest), dim=1)\n"
     "        if self.config.sh_degree > 0:\n"
     "            viewdirs = submodel_means.detach() - camera.camera_to_worlds.detach()[..., :3, 3]  # (N, 3)\n"
     "            viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)\n"
     "            n = min(self.step // self.config.sh_degree_interval, self.config.sh_degree)\n"
     "            if not self.training:\n"
     "                n = self.config.sh_degree\n"
     "            rgbs = spherical_harmonics(n, viewdirs, colors)\n"
     "            rgbs = torch.clamp(rgbs + 0.5, min=0.0)  # type: ignore\n"
     "        else:\n"
     "            rgbs = torch.sigmoid(colors[:, 0, :])\n"
     "        gaussian_attrs = {\n"
     "            \"means\": submodel_means,\n"
     "            \"colors\": colors,\n"
     "            \"opacities\": submodel_opacities,\n"
     "            \"xys\": submodel_xys,\n"
     "            \"depths\": submodel_depths,\n"
     "            \"radii\": submodel_radii,\n"
     "            \"conics\": submodel_conics,\n"
     "            \"num_tiles_hit\": submodel_num_tiles_hit,\n"
     "        }\n"
     "        if sky_capture is not None:\n"
     "            gaussian_attrs[\"sky_capture\"] = sky_capture\n"
     "        outputs = self.render_gaussian_attrs(camera, gaussian_attrs, output_names)\n",
     "        # sgn_fused: the sub-models are a run of consecutive row blocks of the main pass's tensors\n"
     "        assert self.sgn_main_attrs is not None  # (nothing visible: the reference's num_tiles_hit assertion fires here)\n"
     "        first = self.visible_model_names.index(submodel_names[0])\n"
     "        assert self.visible_model_names[first:first + len(submodel_names)] == list(submodel_names)\n"
     "        counts = [self.all_models[name].num_points for name in self.visible_model_names]\n"
     "        id_range = (sum(counts[:first]), sum(counts[:first + len(submodel_names)]))\n"
     "        if output_names == ['accumulation'] and len(self.sgn_group_acc) == 2 and id_range in (\n"
     "                (0, self.sgn_group_split), (self.sgn_group_split, sum(counts))):\n"
     "            # the main pass's own walk accumulated this image (rasterize_gaussians_fused(group_split=...))\n"
     "            camera.rescale_output_resolution(camera_downscale)\n"
     "            return {'accumulation': self.sgn_group_acc[0 if id_range[0] == 0 else 1][..., None]}\n"
     "        gaussian_attrs = {k: v for k, v in self.sgn_main_attrs.items() if k != \"sky_capture\"}\n"
     "        if sky_capture is not None:\n"
     "            gaussian_attrs[\"sky_capture\"] = sky_capture\n"
     "        outputs = self.render_gaussian_attrs(camera, gaussian_attrs, output_names, id_range=id_range)\n"),
    # get_outputs: raw per-object tensors + one row per object in the small per-step tables
    ("        object_features_dc = []\n"
     "        assert camera.times is not None\n",
     "        object_features_dc = []\n"
     "        sgn_poses, sgn_idft = [], []  # per visible object: (rot, center, q_o2w) and the frame's idft weights\n"
     "        assert camera.times is not None\n"),
    ("                if self.config.fourier_features_dim > 1:\n"
     "                    object_features_dc.append(self.get_fourier_features(anno.frame, trackId, obj_model))\n"
     "                else:\n"
     "                    object_features_dc.append(obj_model.features_dc)\n",
     "                object_features_dc.append(obj_model.features_dc)  # sgn_fused: the raw (Fourier) coefficients\n"
     "                sgn_idft.append(self.get_fourier_features(anno.frame, trackId, obj_model, weights_only=True)\n"
     "                                if self.config.fourier_features_dim > 1 else None)\n"),
    ("                obj_means, obj_quats = object2world_gs(\n"
     "                    obj_model.means, obj_model.quats, anno.center, anno.rot)\n"
     "                object_means.append(obj_means)\n"
     "                object_quats.append(obj_quats)\n",
     "                object_means.append(obj_model.means)  # sgn_fused: LOCAL frame; R, t and q_o2w are applied in-kernel\n"
     "                object_quats.append(obj_model.quats)\n"
     "                sgn_poses.append((anno.rot, anno.center, sgn_fused.memo(quaternion_from_matrix, anno.rot)))\n"),
    ("        self.features_dc = torch.cat([self.background_model.features_dc, *object_features_dc], dim=0)\n",
     "        # sgn_fused: the SH coefficients stay one tensor per sub-model (sgn_sh_fwd_parts reads them where they are)\n"
     "        self.features_dc = ((self.background_model.features_dc, *object_features_dc) if self.config.sh_degree > 0\n"
     "                            else sgn_fused.cat_features_dc([self.background_model.features_dc, *object_features_dc]))\n"
     "        self.sgn_tables = sgn_fused.scene_graph_tables(\n"
     "            [self.background_model.num_points] + [m.shape[0] for m in object_means], sgn_poses, sgn_idft, self.device)\n"
     "        # background_acc / object_acc (below) ride on the main pass: ids below / from this split\n"
     "        self.sgn_group_split = self.background_model.num_points if self.training else None\n"),
    ("        self.features_rest = self.get_aggreated_variable(\"features_rest\")\n",
     "        self.features_rest = (tuple(self.all_models[name].features_rest for name in self.visible_model_names)\n"
     "                              if self.config.sh_degree > 0 else self.get_aggreated_variable(\"features_rest\"))\n"),
]


def _apply(src: str, edits) -> str:
    for old, new in edits:
        assert src.count(old) == 1, ("edit does not apply exactly once", old[:70], src.count(old))
        src = src.replace(old, new)
    return src


def patched(src: str) -> str:
    return _apply(src, EDITS)


def patched_scene_graph(src: str) -> str:
    return _apply(src, EDITS_SG)


def _diff(rel, new_fn):
    src = open(os.path.join(REF, rel)).read()
    return "".join(difflib.unified_diff(src.splitlines(True), new_fn(src).splitlines(True), "a/" + rel, "b/" + rel, n=2))


def main():
    for name, rel, fn, n in (("fused_callsites.patch", REL, patched, len(EDITS)),
                             ("fused_scene_graph.patch", REL_SG, patched_scene_graph, len(EDITS_SG))):
        diff = _diff(rel, fn)
        open(os.path.join(HERE, name), "w").write(diff)
        print(f"{name}: {diff.count(chr(10))} lines of patch, {n} edits")


if __name__ == "__main__":
    main()
