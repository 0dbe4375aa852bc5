#!/usr/bin/env python
"""Generates tests/golden/*.npz|json by running the REFERENCE's own Python composition
(/root/reference/pcdet/ops/pointnet2/pointnet2_batch/{pointnet2_utils,pointnet2_modules,
PointFormer}.py and pcdet/models/backbones_3d/IASSD_backbone.py) on CPU, with its CUDA
extension module replaced by a stub backed by this repo's CPU oracle.

Run here only (the GPU box has no /root/reference):  python tests/golden/make_golden.py
The reference sources are imported from where they lie; nothing of them is copied.  What is
committed is data: seeds, small inputs, and the reference composition's outputs.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import oracle  # noqa: E402
from detweights import fill_deterministic  # noqa: E402

torch.set_num_threads(8)


# ---- stub extension: reference-named entry points on CPU tensors, backed by the oracle ----
def _np(t):
    assert t.device.type == "cpu" and t.is_contiguous()
    return t.numpy()


stub = types.ModuleType("pointnet2_batch_cuda")
for _name in ["ball_query_wrapper", "ball_query_dilated_wrapper", "group_points_wrapper",
              "group_points_grad_wrapper", "gather_points_wrapper", "gather_points_grad_wrapper",
              "farthest_point_sampling_wrapper", "furthest_point_sampling_with_dist_wrapper",
              "three_nn_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper"]:
    def _mk(name):
        fn = getattr(oracle, name)

        def call(*args):
            return fn(*[_np(a) if isinstance(a, torch.Tensor) else a for a in args])
        return call
    setattr(stub, _name, _mk(_name))


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def import_reference():
    # the reference allocates with torch.cuda.IntTensor/FloatTensor (pointnet2_utils.py:25-26...)
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    sys.modules["open3d"] = types.ModuleType("open3d")  # semantic_view.py:1 (visualisation only)
    base = REF + "/pcdet/ops/pointnet2/pointnet2_batch"
    _pkg("pcdet", REF + "/pcdet")
    _pkg("pcdet.ops", REF + "/pcdet/ops")
    _pkg("pcdet.ops.pointnet2", REF + "/pcdet/ops/pointnet2")
    _pkg("pcdet.ops.pointnet2.pointnet2_batch", base)
    sys.modules["pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda"] = stub
    sys.modules["pcdet.ops.pointnet2.pointnet2_batch"].pointnet2_batch_cuda = stub

    def load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod
    pu = load("pcdet.ops.pointnet2.pointnet2_batch.pointnet2_utils", base + "/pointnet2_utils.py")
    load("pcdet.ops.pointnet2.pointnet2_batch.PointFormer", base + "/PointFormer.py")
    load("pcdet.ops.pointnet2.pointnet2_batch.semantic_view", base + "/semantic_view.py") \
        if False else sys.modules.setdefault("pcdet.ops.pointnet2.pointnet2_batch.semantic_view",
                                             types.ModuleType("semantic_view"))
    pm = load("pcdet.ops.pointnet2.pointnet2_batch.pointnet2_modules", base + "/pointnet2_modules.py")
    # backbone: parents + the unused torchsparse import (IASSD_backbone.py:7)
    _pkg("pcdet.models", REF + "/pcdet/models")
    _pkg("pcdet.models.backbones_3d", REF + "/pcdet/models/backbones_3d")
    _pkg("pcdet.models.backbones_3d.cluster")
    sp = types.ModuleType("pcdet.models.backbones_3d.cluster.spvnas_cluster")
    sp.SPVNAS = object
    sys.modules["pcdet.models.backbones_3d.cluster.spvnas_cluster"] = sp
    bb = load("pcdet.models.backbones_3d.IASSD_backbone", REF + "/pcdet/models/backbones_3d/IASSD_backbone.py")
    return pu, pm, bb


class AD(dict):
    def __getattr__(self, k):
        return self[k]


def to_ad(o):
    if isinstance(o, dict):
        return AD({k: to_ad(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_ad(v) for v in o]
    return o


def scene_points(b, n, seed):
    from pdanet_amd import synth
    return synth.batch_points(b, n, config_id=seed, dist="L")


def t2n(x):
    return x.detach().cpu().numpy()


def main():
    pu, pm, bb = import_reference()
    out = {}
    meta = {"generator": "tests/golden/make_golden.py", "reference": "Geo3DSmart/PDANet @ /root/reference",
            "torch": torch.__version__}

    # 1. state-dict schema of the reference backbone for both yamls
    schema = {}
    for tag, path in [("once", REF + "/tools/cfgs/once_models/PDA-SSD.yaml"),
                      ("kitti", REF + "/tools/cfgs/kitti_models/PDA-SSD.yaml")]:
        cfg = to_ad(yaml.safe_load(open(path)))
        model = bb.IASSD_Backbone(cfg.MODEL.BACKBONE_3D, num_class=len(cfg.CLASS_NAMES), input_channels=4)
        schema[tag] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        meta[tag + "_params"] = sum(p.numel() for p in model.parameters())
    json.dump(schema, open(os.path.join(HERE, "backbone_state_dict_schema.json"), "w"))

    # 2. pure-torch sub-blocks
    torch.manual_seed(0)
    PF = sys.modules["pcdet.ops.pointnet2.pointnet2_batch.PointFormer"]
    tr = fill_deterministic(PF.TransformerEncoderLayerPreNorm(d_model=32, nhead=4, dim_feedforward=16, dropout=0.0)).eval()
    x = torch.randn(8, 6, 32)
    out["transformer_in"] = t2n(x); out["transformer_out"] = t2n(tr(x))
    dn = fill_deterministic(pm.PointConvDensitySetAbstraction(0.8))
    d = torch.rand(2, 1, 5, 8) + 0.1
    dn.eval(); out["density_in"] = t2n(d); out["density_out_eval"] = t2n(dn(d))
    dn.train(); out["density_out_train"] = t2n(dn(d))
    vote = fill_deterministic(pm.Vote_layer(mlp_list=[16], pre_channel=8, max_translate_range=[3.0, 3.0, 2.0])).eval()
    vx, vf = torch.randn(2, 10, 3) * 5, torch.randn(2, 8, 10) * 3
    v = vote(vx, vf)
    out["vote_xyz_in"], out["vote_feat_in"] = t2n(vx), t2n(vf)
    out["vote_xyz"], out["vote_offsets"] = t2n(v[0]), t2n(v[3])

    # 3. one vanilla SA layer and one PDA layer through the stub extension (B=2, N=1024)
    pts = scene_points(2, 1024, seed=77)
    xyz = torch.from_numpy(pts[:, 1:4]).view(2, 1024, 3).contiguous()
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2, 6, 1024, generator=g)
    cls_feats = torch.randn(2, 1024, 3, generator=g)
    out["sa_xyz"], out["sa_feats"], out["sa_cls"] = t2n(xyz), t2n(feats), t2n(cls_feats)
    sa_kwargs = dict(npoint_list=[256], sample_range_list=[-1], sample_type_list=["D-FPS"],
                     radii=[2.0, 6.0], nsamples=[8, 16], use_xyz=True, dilated_group=False,
                     aggregation_mlp=[24], confidence_mlp=[12], num_class=3)
    meta["sa_kwargs"] = sa_kwargs
    meta["sa_mlps"] = [[6, 8, 16], [6, 8, 12]]
    meta["pda_mlps"] = [[6, 8, 16], [6, 10, 12]]
    for name, cls, mlps in [("sa", pm.PointnetSAModuleMSG_WithSampling, meta["sa_mlps"]),
                            ("pda", pm.PointnetSAModuleMSG_WithSampling_Ellipsoid, meta["pda_mlps"])]:
        if name == "pda":
            mlps = [[8] + m[1:] for m in mlps]  # C = 8: feature channels must equal mlp_spec[0]
            meta["pda_mlps"] = mlps
            f_in = torch.randn(2, 8, 1024, generator=g)
            out["pda_feats"] = t2n(f_in)
        else:
            f_in = feats
        layer = fill_deterministic(cls(mlps=[list(m) for m in mlps], **sa_kwargs))
        for mode in ["eval", "train"]:
            layer.train(mode == "train")
            with torch.no_grad():
                nx, nf, cf, sidx = layer(xyz, f_in, None)
            out["%s_%s_new_xyz" % (name, mode)] = t2n(nx)
            out["%s_%s_new_features" % (name, mode)] = t2n(nf)
            out["%s_%s_cls" % (name, mode)] = t2n(cf)
            out["%s_%s_idx" % (name, mode)] = t2n(sidx)
        # ctr-aware sampling variant (top-k of sigmoid(max cls))
        kw = dict(sa_kwargs); kw["sample_type_list"] = ["ctr_aware"]
        layer2 = fill_deterministic(cls(mlps=[list(m) for m in mlps], **kw)).eval()
        with torch.no_grad():
            nx, nf, cf, sidx = layer2(xyz, f_in, cls_feats)
        out["%s_ctr_new_features" % name] = t2n(nf)
        out["%s_ctr_idx" % name] = t2n(sidx)
    # vanilla layer centred on given ctr_xyz (layer-5 style: centres are not points)
    ctr = xyz[:, :64].contiguous() + 0.3
    layer = fill_deterministic(pm.PointnetSAModuleMSG_WithSampling(mlps=[list(m) for m in meta["sa_mlps"]], **sa_kwargs)).eval()
    with torch.no_grad():
        nx, nf, cf, sidx = layer(xyz, feats, None, ctr_xyz=ctr)
    out["sa_ctrxyz_in"] = t2n(ctr); out["sa_ctrxyz_new_features"] = t2n(nf)

    # 4. groupers on their own
    gq = pu.QueryAndGroup(2.0, 8)(xyz, ctr, feats)
    gd = pu.QueryAndGroup_alone_grouped_density_directional(2.0, 8)(xyz, ctr, feats)
    out["grouper_vanilla"], out["grouper_pda"] = t2n(gq), t2n(gd)

    # 5. FP module (three_nn + three_interpolate composition)
    fp = fill_deterministic(pm.PointnetFPModule(mlp=[6 + 4, 16])).eval()
    known = xyz[:, :100].contiguous(); kf = torch.randn(2, 6, 100, generator=g); uf = torch.randn(2, 4, 1024, generator=g)
    with torch.no_grad():
        out["fp_out"] = t2n(fp(xyz, known, uf, kf))
    out["fp_known_feats"], out["fp_unknown_feats"] = t2n(kf), t2n(uf)

    # 6. the whole backbone, scaled-down point counts (real channel widths), eval + train BN
    cfg = to_ad(yaml.safe_load(open(REF + "/tools/cfgs/once_models/PDA-SSD.yaml")))
    sa = cfg.MODEL.BACKBONE_3D.SA_CONFIG
    sa["NPOINT_LIST"] = [[2048], [512], [256], [128], [-1], [128]]
    meta["backbone_npoint_list"] = sa["NPOINT_LIST"]
    model = fill_deterministic(bb.IASSD_Backbone(cfg.MODEL.BACKBONE_3D, num_class=5, input_channels=4))
    bpts = torch.from_numpy(scene_points(2, 2048, seed=91))
    for mode in ["eval", "train"]:
        model.train(mode == "train")
        with torch.no_grad():
            bd = model({"batch_size": 2, "points": bpts.clone()})
        for k in ["centers", "centers_origin", "ctr_offsets", "centers_features"]:
            out["bb_%s_%s" % (mode, k)] = t2n(bd[k])
        for li, t in enumerate(bd["encoder_xyz"]):
            out["bb_%s_encoder_xyz_%d" % (mode, li)] = t2n(t)
        for li, t in enumerate(bd["sa_ins_preds"]):
            if not isinstance(t, list):
                out["bb_%s_sa_ins_preds_%d" % (mode, li)] = t2n(t)
        for li, t in enumerate(bd["encoder_features"][1:], start=1):
            # keep fixtures small: per-layer feature checksums + a slice
            out["bb_%s_feat_%d_slice" % (mode, li)] = t2n(t[:, :8, :16])
            out["bb_%s_feat_%d_absmean" % (mode, li)] = np.array([t.abs().mean().item()], np.float32)

    np.savez_compressed(os.path.join(HERE, "reference_composition.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "reference_composition_meta.json"), "w"), indent=1)
    print("wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
