#!/usr/bin/env python
"""Close the one unverified part of row a3: do the backbone parameter names of this build equal timm 0.6.7's?

``gdrnpp_bop2022_amd/gdrn_modeling/backbones.py`` re-declares ``timm.create_model("convnext_base" | "resnet34",
features_only=True, out_indices=...)`` (reference: core/utils/timm_utils.py:34, models/net_factory.py:73-74) from timm's published
source, because timm is not installable where this library was written.  A maintainer who HAS timm 0.6.7 and/or a GDRNPP
checkpoint runs this script once; it needs no GPU.

  python tools/check_timm_keys.py --timm                       # diff against timm.create_model(..., features_only=True)
  python tools/check_timm_keys.py --checkpoint model_final.pth  # diff against the ``backbone.*`` keys of a GDRNPP checkpoint
  python tools/check_timm_keys.py --timm --config lmo_resnet34_ape
  python tools/check_timm_keys.py --manifest                    # (no timm) print this build's keys as JSON, for a manual diff

Exit status 0 = every key and shape agrees (then ``load_checkpoint(model, path, strict=True)`` loads the backbone by name);
1 = a difference, listed as  missing-here / extra-here / shape-mismatch.  With --checkpoint the whole file is also loaded with
``GDRN_double_mask.load_checkpoint(strict=True)`` and, with --forward, one CPU forward on a seeded image is compared between this
build's module and timm's (same state_dict, max abs difference of the feature map printed)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def our_backbone(cfg_name):
    import torch

    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.backbones import create_backbone

    cfg = get_cfg(cfg_name, opts=["MODEL.DEVICE=cpu"])
    init = dict(cfg.MODEL.POSE_NET.BACKBONE.INIT_CFG)
    torch.manual_seed(0)
    return cfg, init, create_backbone(**init)


def diff(ours: dict, theirs: dict, what: str) -> int:
    missing = sorted(k for k in theirs if k not in ours)
    extra = sorted(k for k in ours if k not in theirs)
    shape = sorted(k for k in ours if k in theirs and tuple(ours[k]) != tuple(theirs[k]))
    print(f"== {what}: {len(theirs)} keys there, {len(ours)} here")
    for title, ks in (("missing here (in the other, not in this build)", missing), ("extra here", extra)):
        if ks:
            print(f"  {title}: {len(ks)}")
            for k in ks[:20]:
                print("     ", k, theirs.get(k, ours.get(k)))
    if shape:
        print(f"  shape mismatch: {len(shape)}")
        for k in shape[:20]:
            print("     ", k, "here", tuple(ours[k]), "there", tuple(theirs[k]))
    bad = len(missing) + len(extra) + len(shape)
    print("  ->", "IDENTICAL key set and shapes" if bad == 0 else f"{bad} differences")
    return bad


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default="ycbv_convnext_a6", help="named config of this build (its BACKBONE.INIT_CFG is used)")
    ap.add_argument("--timm", action="store_true", help="compare with timm.create_model(features_only=True) (needs timm, ideally 0.6.7)")
    ap.add_argument("--checkpoint", help="a GDRNPP checkpoint ({'model': state_dict} or a bare state_dict)")
    ap.add_argument("--forward", action="store_true", help="with --timm: also compare one CPU forward on the same parameters")
    ap.add_argument("--manifest", action="store_true", help="print this build's backbone keys/shapes as JSON and exit")
    args = ap.parse_args()
    import torch

    cfg, init, bb = our_backbone(args.config)
    ours = {k: tuple(v.shape) for k, v in bb.state_dict().items()}
    if args.manifest:
        print(json.dumps({k: list(v) for k, v in ours.items()}, indent=0))
        return 0
    bad = 0
    if args.timm:
        import timm

        name = init["type"].split("/")[-1]
        kw = {k: v for k, v in init.items() if k not in ("type", "pretrained")}
        ref = timm.create_model(model_name=name, pretrained=False, **kw)
        print(f"timm {timm.__version__}: create_model({name!r}, {kw})")
        theirs = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        bad += diff(ours, theirs, f"timm {timm.__version__} {name}")
        if args.forward and bad == 0:
            from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
            hip_layers.set_enabled(False)
            ref.load_state_dict(bb.state_dict(), strict=True)
            ref.eval(), bb.eval()
            x = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(1))
            with torch.no_grad():
                a, b = bb(x)[-1], ref(x)[-1]
            err = (a - b).abs().max().item() / b.abs().max().item()
            print(f"  forward on shared parameters: max |ours - timm| / max |timm| = {err:.2e}")
            bad += int(err > 1e-5)
    if args.checkpoint:
        sd = torch.load(args.checkpoint, map_location="cpu")
        sd = sd.get("model", sd)
        strip = lambda k: k[len("module."):] if k.startswith("module.") else (k[len("_module."):] if k.startswith("_module.") else k)  # noqa: E731
        theirs = {strip(k)[len("backbone."):]: tuple(v.shape) for k, v in sd.items() if strip(k).startswith("backbone.")
                  and not k.endswith("num_batches_tracked")}
        ours_ck = {k: v for k, v in ours.items() if not k.endswith("num_batches_tracked")}
        bad += diff(ours_ck, theirs, f"checkpoint {os.path.basename(args.checkpoint)} (backbone.*)")
        from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer, load_checkpoint
        model, _ = build_model_optimizer(cfg)
        try:
            load_checkpoint(model, args.checkpoint, strict=True)
            print("  load_checkpoint(strict=True): OK — every parameter of the model was found in the file")
        except RuntimeError as e:
            print("  load_checkpoint(strict=True) FAILED:", e)
            bad += 1
    if not (args.timm or args.checkpoint):
        ap.error("give --timm and/or --checkpoint (or --manifest)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
