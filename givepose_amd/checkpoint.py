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


def load_checkpoint(net, state_dict, strict=True, verbose=True):
    """`net.load_state_dict` through `remap_keys`; like evaluate.py:53-56 the checkpoint may be partial (missing keys keep
    the module's current values), unknown keys are an error under `strict`."""
    sd, renamed = remap_keys(state_dict)
    own = net.state_dict()
    unknown = [k for k in sd if k not in own]
    if unknown and strict:
        raise KeyError(f"checkpoint keys that match no tensor of {type(net).__name__}: {unknown[:8]}{' ...' if len(unknown) > 8 else ''}")
    bad = [(k, tuple(sd[k].shape), tuple(own[k].shape)) for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError(f"shape mismatch: {bad[:4]}")
    merged = dict(own)
    merged.update({k: torch.as_tensor(v) for k, v in sd.items() if k in own})
    net.load_state_dict(merged, strict=True)
    if verbose and renamed:
        print(f"[givepose_amd] {len(renamed)} checkpoint keys renamed, e.g. {next(iter(renamed.items()))}")
    return {"renamed": renamed, "missing": [k for k in own if k not in sd], "unknown": unknown}
