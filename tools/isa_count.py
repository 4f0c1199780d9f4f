"""Static instruction counts of the pairing unit's device functions (gfx950 ISA from hipcc -S): VALU / MAD / DPP / select / scratch per
function, and for f12p_expt_to its inner loops (the cyclotomic-squaring run and the window product).  Kernel-tuning aid: the Fp12
kernels are issue-bound, so time follows the executed VALU count.
usage: python tools/isa_count.py [csrc dir] [-- extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm(csrc, flags=()):
    out = os.path.join(tempfile.mkdtemp(prefix="gpbc_isa_"), "pairing.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), *flags,
                           os.path.join(csrc, "gpbc_pairing.hip"), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def classify(op):
    if op.startswith("v_mad_i64") or op.startswith("v_mad_u64"): return "mad"
    if "dpp" in op: return "dpp"
    if op.startswith("v_cndmask"): return "sel"
    if op.startswith("v_mov"): return "mov"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("s_"): return "salu"
    return "other"


def count(lines):
    c = collections.Counter()
    for ln in lines:
        m = re.match(r"\s+([a-z_0-9]+)", ln)
        if m and not ln.lstrip().startswith((".", ";")):
            c[classify(m.group(1))] += 1
    c["valu"] = c["mad"] + c["dpp"] + c["sel"] + c["mov"] + c["valu_other"]
    return c


def functions(lines):
    fn, start = None, 0
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s+; @", ln)
        if m:
            fn, start = m.group(1), i
        elif fn and ln.startswith(".Lfunc_end"):
            yield fn, lines[start:i]
            fn = None


def main():
    args = sys.argv[1:]
    flags = []
    if "--" in args:
        flags = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    csrc = args[0] if args else os.path.join(ROOT, "gopairingbasedcryptography_amd", "csrc")
    lines = asm(csrc, flags)
    want = ("f12p_expt_to", "f2_mul_leaf", "f2_sqr_leaf", "fe_mul_leaf", "k_final_exp", "k_miller_accumulateP", "k_miller_linesP")
    for fn, body in functions(lines):
        if not any(w in fn for w in want): continue
        c = count(body)
        print("%-70s valu %6d  mad %5d  dpp %4d  sel %4d  mov %4d  scratch %4d  lds %3d  vmem %3d" % (fn[:70], c["valu"], c["mad"], c["dpp"], c["sel"], c["mov"], c["scratch"], c["lds"], c["vmem"]))
        if "f12p_expt_to" in fn or "k_miller_accumulateP" in fn:
            # inner loops: blocks between a loop header label and its backward branch
            labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
            for i, ln in enumerate(body):
                m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
                if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > 300:
                    blk = body[labels[m.group(1)]:i]
                    cc = count(blk)
                    calls = sum(1 for b in blk if "s_swappc" in b)
                    print("    loop %s (%d lines, %d leaf calls): valu %5d  mad %4d  dpp %4d  sel %4d  mov %4d  scratch %3d" % (m.group(1), len(blk), calls, cc["valu"], cc["mad"], cc["dpp"], cc["sel"], cc["mov"], cc["scratch"]))
    for ln in lines:
        if re.search(r"\.(vgpr_spill_count|private_segment_fixed_size|vgpr_count):", ln) or ".name:" in ln and "k_" in ln:
            pass
    # kernel resource summary from the metadata
    name = None
    for ln in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m: name = m.group(1)
        m = re.match(r"\s+\.(private_segment_fixed_size|vgpr_spill_count|vgpr_count):\s+(\d+)", ln)
        if m and name and any(k in name for k in ("k_final_exp", "k_miller_accumulateP", "k_miller_linesP")):
            print("  %-40s %-28s %s" % (name[:40], m.group(1), m.group(2)))


if __name__ == "__main__":
    main()
