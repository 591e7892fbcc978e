"""What the range guard judged on a TRAINED network: per absmax block of the trained-weights parity cases --
the largest magnitude, how many of the block's non-zero words lie within 2^-16 of it, and how many the outlier rule needs
(conv_guard.h: code 1 fires below `need`) -- plus the hand-over's loose bits and the depth error budget.

    python scripts/trained_guard_stats.py > profiles/r06_trained_guard_stats.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def block_stats(block):
    w = block.cpu().to(torch.int64) & 0xffffffff
    m = int(w.max())
    nz = int((w != 0).sum())
    thr = m - (16 << 23) if m > (16 << 23) else 1
    near = int((w >= thr).sum())
    need = min(nz, max(2, nz >> 3))
    val = torch.tensor([m], dtype=torch.int64).to(torch.int32).view(torch.float32).item() if m < 2 ** 31 else float("nan")
    return {"max": val, "nonzero_words": nz, "within_2^-16": near, "needed": need, "margin_words": near - need,
            "code": 2 if m >= 0x7f800000 else (1 if near < need else 0)}


def main():
    from mvs_amd import ops
    from trained_cases import run_trained
    out = {}
    for which in ("small", "full"):
        t = ops.StageTimer()
        ops.set_timer(t)          # the per-layer chain: its blocks stay readable
        try:
            r = run_trained(which, True, keep_model=True)
        finally:
            ops.set_timer(None)
        model = r.pop("_model")
        r.pop("_case")
        torch.cuda.synchronize()
        names = ["conv0", "conv1", "conv3", "conv5", "conv6", "conv7", "conv9", "conv2", "conv4"]
        cb = model.cost_regularization._last_blocks
        fb = model.feature._last_blocks
        r["costreg_blocks"] = {n: block_stats(cb[i]) for i, n in enumerate(names)}
        r["feature_blocks"] = {f"row{i}": block_stats(fb[i]) for i in range(fb.shape[0]) if int((fb[i] != 0).sum())}
        hw = getattr(model, "_last_handover_words", None)
        if hw is not None:
            hand, amax, redo = hw
            r["handover"] = {"redo": int(redo[0].item()), "bound": block_stats(hand)["max"], "variance_block": block_stats(amax),
                             "loose_bits": (int(hand.max().item()) >> 23) - (int(amax.max().item()) >> 23)}
        out[which] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
