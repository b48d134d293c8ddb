"""Checkpoint key handling for `PoseNet.load_state_dict` (evaluation/evaluate.py:53-56 loads the released
weights with `model_dict.update(torch.load(FLAGS.resume_model)); network.load_state_dict(model_dict)`).

The non-backbone names are the reference's own (tests/golden/state_dict_manifest.json).  `backbone.*` comes from timm 0.9.6
`create_model("convnext_base", features_only=True)` = `FeatureListNet`, which flattens the module tree with underscores
(`stem_0`, `stages_1.blocks.0.conv_dw`, ...).  timm is not installed here and no released checkpoint is reachable, so
that layout is from memory; `remap_keys` therefore also accepts the spellings a checkpoint could plausibly carry instead
-- the un-flattened timm ConvNeXt names (`stem.0`, `stages.1.blocks.0`), the HuggingFace `ConvNextModel` names the
cross-check uses, a DataParallel `module.` prefix -- and maps them onto the names this class registers.  Nothing is
guessed silently: `load_checkpoint` reports what it renamed and raises on anything it cannot place (strict).
"""
import re

import torch

from . import synth


def remap_keys(state_dict):
    """-> (remapped dict, {old name: new name} of every key that was renamed)."""
    out, renamed = {}, {}
    for k, v in state_dict.items():
        n = k[7:] if k.startswith("module.") else k
        if n.startswith("backbone."):
            b = n[len("backbone."):]
            b = re.sub(r"^stem\.(\d)\.", r"stem_\1.", b)                       # timm ConvNeXt (not FeatureListNet)
            b = re.sub(r"^stages\.(\d)\.", r"stages_\1.", b)
            hf = synth.hf_to_timm(b) if b.startswith(("embeddings.", "encoder.")) else None   # HuggingFace ConvNextModel
            if hf is not None:
                b = hf[len("backbone."):] if hf.startswith("backbone.") else hf
            n = "backbone." + b
        if n != k:
            renamed[k] = n
        out[n] = v
    return out, renamed


def write_report(path, report, state_dict=None, own=None):
    """The full key report of a load as one JSON file: every rename, every checkpoint key that matched nothing (with its shape),
    every tensor of the module the checkpoint did not provide, every shape mismatch.  The timm `FeatureListNet` key layout of
    `backbone.*` is unverified here (timm is not installed, no released checkpoint is reachable offline): the first user who loads
    the released weights (README.md:55 of the reference) can send back THIS file and it settles the layout."""
    import json
    shp = lambda d, k: (list(d[k].shape) if d is not None and k in d and hasattr(d[k], "shape") else None)
    doc = {"what": "givepose_amd.checkpoint.load_checkpoint key report",
           "checkpoint_keys": None if state_dict is None else len(state_dict), "module_keys": None if own is None else len(own),
           "renamed": [{"checkpoint": a, "module": b, "shape": shp(state_dict, a)} for a, b in report["renamed"].items()],
           "unknown": [{"checkpoint": k, "shape": shp(report.get("_sd"), k)} for k in report["unknown"]],
           "missing": [{"module": k, "shape": shp(own, k)} for k in report["missing"]],
           "shape_mismatch": [{"key": k, "checkpoint": list(a), "module": list(b)} for k, a, b in report.get("shape_mismatch", [])]}
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    return path


def load_checkpoint(net, state_dict, strict=True, verbose=True, report_path=None):
    """`net.load_state_dict` through `remap_keys`; like evaluate.py:53-56 the checkpoint may be partial (missing keys keep
    the module's current values), unknown keys are an error under `strict`.
    report_path (or the environment variable GP_CHECKPOINT_REPORT): write the FULL rename / unknown / missing / shape-mismatch report
    there as JSON -- also, and especially, when the load is about to fail (write_report)."""
    import os
    report_path = report_path or os.environ.get("GP_CHECKPOINT_REPORT")
    sd, renamed = remap_keys(state_dict)
    own = net.state_dict()
    unknown = [k for k in sd if k not in own]
    if report_path:
        mism = [(k, tuple(sd[k].shape), tuple(own[k].shape)) for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
        write_report(report_path, {"renamed": renamed, "unknown": unknown, "missing": [k for k in own if k not in sd], "shape_mismatch": mism, "_sd": sd},
                     state_dict, own)
        if verbose:
            print(f"[givepose_amd] checkpoint key report: {report_path} ({len(renamed)} renamed, {len(unknown)} unknown, {len(mism)} shape mismatches)")
    if unknown and strict:
        raise KeyError(f"checkpoint keys that match no tensor of {type(net).__name__}: {unknown[:8]}{' ...' if len(unknown) > 8 else ''}"
                       + ("" if report_path else "  (set GP_CHECKPOINT_REPORT=<file> or report_path= for the full key report)"))
    bad = [(k, tuple(sd[k].shape), tuple(own[k].shape)) for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError(f"shape mismatch: {bad[:4]}")
    merged = dict(own)
    merged.update({k: torch.as_tensor(v) for k, v in sd.items() if k in own})
    net.load_state_dict(merged, strict=True)
    if verbose and renamed:
        print(f"[givepose_amd] {len(renamed)} checkpoint keys renamed, e.g. {next(iter(renamed.items()))}")
    return {"renamed": renamed, "missing": [k for k in own if k not in sd], "unknown": unknown}


EXPECTED_KEYS_FILE = "tests/golden/expected_checkpoint_keys.txt"


def expected_keys(cfg=None):
    """[(key, shape)] of everything `PoseNet(cfg)` registers, in registration order: what a released checkpoint has to provide (or a
    subset of it: evaluate.py:53-56 updates the model's own dict).  `backbone.*` follows the timm 0.9.6 `FeatureListNet` flattening as
    this package understands it -- the one part no fixture of the reference pins (module docstring)."""
    from .config import PoseNetConfig
    from .posenet import PoseNet
    net = PoseNet(cfg or PoseNetConfig(), dtype=torch.float32, seed=0)
    return [(k, tuple(v.shape)) for k, v in net.state_dict().items()]


def diff_keys(state_dict, expected):
    """A checkpoint's keys against the expected list (after `remap_keys`): -> dict(renamed, unknown, missing, shape_mismatch)."""
    sd, renamed = remap_keys(state_dict)
    exp = dict(expected)
    return {"renamed": renamed,
            "unknown": [(k, tuple(getattr(v, "shape", ()))) for k, v in sd.items() if k not in exp],
            "missing": [(k, s) for k, s in expected if k not in sd],
            "shape_mismatch": [(k, tuple(sd[k].shape), exp[k]) for k in sd if k in exp and tuple(sd[k].shape) != tuple(exp[k])]}


def _read_expected(path):
    out = []
    for line in open(path):
        if line.strip() and not line.startswith("#"):
            k, s = line.rstrip("\n").split("\t")
            out.append((k, tuple(int(x) for x in s.split("x")) if s != "scalar" else ()))
    return out


def main(argv=None):
    """python -m givepose_amd.checkpoint <checkpoint.pth> [expected_keys.txt]   -- one command for a user who HAS the released weights
    (README.md:55 of the reference; none is reachable offline): prints what the checkpoint calls differently from the committed list
    (tests/golden/expected_checkpoint_keys.txt), what it carries that nothing expects and what it lacks.  CPU only.
    python -m givepose_amd.checkpoint --write <expected_keys.txt>            -- regenerate the list from the module."""
    import os
    import sys
    argv = sys.argv[1:] if argv is None else argv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if argv and argv[0] == "--write":
        path = argv[1] if len(argv) > 1 else os.path.join(root, EXPECTED_KEYS_FILE)
        with open(path, "w") as f:
            f.write("# every tensor PoseNet(PoseNetConfig()) registers: <key>\\t<shape>; backbone.* = timm 0.9.6 FeatureListNet(convnext_base) names as this package\n"
                    "# understands them (UNVERIFIED against timm: not installed, no released checkpoint reachable).  python -m givepose_amd.checkpoint <ckpt> diffs a checkpoint against it.\n")
            for k, s in expected_keys():
                f.write(k + "\t" + ("x".join(str(d) for d in s) if s else "scalar") + "\n")
        print("wrote", path)
        return 0
    if not argv:
        print(main.__doc__)
        return 2
    exp = _read_expected(argv[1] if len(argv) > 1 else os.path.join(root, EXPECTED_KEYS_FILE))
    try:        # third-party checkpoints: never unpickle arbitrary objects just to diff key names
        ck = torch.load(argv[0], map_location="cpu", weights_only=True)
    except Exception as e:
        print(f"cannot read {argv[0]} with weights_only=True ({type(e).__name__}: {e}).\n"
              "If you trust the file, re-save its tensors only:  torch.save({k: v for k, v in torch.load(p, weights_only=False)['state_dict'].items()}, out)")
        return 2
    for key in ("state_dict", "model", "network"):       # common wrappers
        if isinstance(ck, dict) and key in ck and isinstance(ck[key], dict):
            ck = ck[key]
    d = diff_keys(ck, exp)
    print(f"{len(ck)} checkpoint tensors, {len(exp)} expected; {len(d['renamed'])} renamed, {len(d['unknown'])} unknown, {len(d['missing'])} missing, "
          f"{len(d['shape_mismatch'])} shape mismatches")
    for a, b in list(d["renamed"].items())[:20]:
        print("  renamed ", a, "->", b)
    for k, s in d["unknown"]:
        print("  UNKNOWN ", k, s)
    for k, s in d["missing"][:12]:
        print("  missing ", k, s)
    if len(d["missing"]) > 12:
        print(f"  ... and {len(d['missing']) - 12} more missing (a partial checkpoint is fine: evaluate.py:53-56 updates the model's own dict)")
    for k, a, b in d["shape_mismatch"]:
        print("  SHAPE   ", k, "checkpoint", a, "expected", b)
    return 1 if d["unknown"] or d["shape_mismatch"] else 0


if __name__ == "__main__":
    raise SystemExit(main())
