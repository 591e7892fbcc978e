"""Static instruction mix of the kernels of one source file (no GPU needed).

    python scripts/isa_count.py sweep variance_bwd_dma_kernelILi2E [--git REV]

Compiles mvs_amd/csrc/<name>.hip to gfx950 assembly with the library's flags (or the file as of a git revision) and prints, per
kernel whose mangled name contains the pattern, the number of vector / scalar / LDS / memory instructions, v_readlane /
v_writelane (SGPR spills), registers and scratch.  Counts are static (every instruction once, loops not weighted): a tool for
before / after comparisons of one kernel, not a timing model.
"""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_amd import build as B

def analyze(name, pat, rev=None):
    """-> ({kernel: Counter of instruction classes}, {kernel: resource metadata}) for the kernels of mvs_amd/csrc/<name>.hip whose
    mangled name contains `pat`; rev: the file as of that git revision."""
    src = os.path.join(B.CSRC, name + ".hip")
    if rev:
        txt = subprocess.check_output(["git", "show", "%s:mvs_amd/csrc/%s.hip" % (rev, name)], cwd=os.path.dirname(B.HERE))
        src = os.path.join(B.CSRC, "_isa_tmp_%s.hip" % name)
        open(src, "wb").write(txt)
    out = tempfile.mktemp(suffix=".s")
    try:
        subprocess.check_call([B.hipcc()] + B.FLAGS + ["--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    finally:
        if rev:
            os.remove(src)
    cur, stats, meta = None, collections.OrderedDict(), {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if pat in m.group(1) else None
            if cur:
                stats[cur] = collections.Counter()
            continue
        if cur is None:
            m = re.match(r"\s+\.(sgpr_count|vgpr_count|agpr_count|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)", line)
            if m and meta.get("_k"):
                meta[meta["_k"]][m.group(1)] = int(m.group(2))
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                meta["_k"] = m.group(1) if pat in m.group(1) else None
                if meta["_k"]:
                    meta[meta["_k"]] = {}
            continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";", "/")) or t[0].endswith(":"):
            if line.startswith("\t.end_amdhsa_kernel") or ".size" in line:
                pass
            continue
        op = t[0]
        c = stats[cur]
        if op.startswith("s_endpgm"):
            c["_end"] += 1
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): c[op.split("_b32")[0]] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if "f64" in op: c["valu_f64"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
            if "add" in op: c["lds_atomic"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): c["vmem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if op.startswith("s_waitcnt"): c["waitcnt"] += 1
            if op.startswith("s_barrier"): c["barrier"] += 1
    os.remove(out)
    meta.pop("_k", None)
    return stats, meta


def main():
    rev = sys.argv[sys.argv.index("--git") + 1] if "--git" in sys.argv else None
    stats, meta = analyze(sys.argv[1], sys.argv[2], rev)
    for k, c in stats.items():
        print(k)
        print("   ", dict(c))
        print("   ", meta.get(k, {}))

if __name__ == "__main__":
    main()
