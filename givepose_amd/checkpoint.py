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
