"""The trained-weights full-size case (tests/trained_cases.py) under the arithmetic switches: how far each build's depth lies from the
float64 answer, beside the reference's own float32 forward.  Each variant in its own process (the switches are read at pack time)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"default": {}, "no_handover": {"MVS_HANDOVER": "0"}, "exact_operands": {"MVS_CONV0_F16": "0", "MVS_SPLIT_F16": "0"},
            "fp32_mfma": {"MVS_CONV_SPLIT": "0"}, "exact_coordinates": {"EXP_FAST": "0"}}


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from trained_cases import run_trained
    which = os.environ.get("EXP_CASE", "full")
    r = run_trained(which, os.environ.get("EXP_FAST", "1") == "1")
    r.pop("conf")
    print("RESULT " + json.dumps(r), flush=True)


def main():
    out = {}
    for case in ("full", "small"):
        for k, env in VARIANTS.items():
            e = dict(os.environ, EXP_CASE=case, **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True, timeout=900)
            line = next((l for l in p.stdout.splitlines() if l.startswith("RESULT ")), None)
            out[f"{case}:{k}"] = json.loads(line[7:]) if line else {"error": (p.stdout + p.stderr)[-800:]}
            print(case, k, json.dumps(out[f"{case}:{k}"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "trained_budget_switches.json"), "w"), indent=1)


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else main()
