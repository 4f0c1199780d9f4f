"""Greedy pruning of carry-free normalisations: drop one fe/f2/f6_norm call at a time and keep the change when the signed-interval
bounds harness still proves every int32 limb and int64 column overflow-free and every result still equals the oracle."""
import re, subprocess, sys, os, time
REPO = "/tmp/prune/repo"
CSRC = os.path.join(REPO, "gopairingbasedcryptography_amd", "csrc")
files = sys.argv[1:]
PAT = re.compile(r"\b(fe|f2|f6)_norm\(")
def run_tests():
    so = os.path.join(REPO, "tools", "libgpbc_bounds.so")
    if os.path.exists(so): os.remove(so)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_device_math_bounds.py", "tests/test_wire.py", "tests/test_hash_to_curve.py", "-x", "-q"],
                       cwd=REPO, capture_output=True, text=True, timeout=1500)
    return r.returncode == 0
log = open("/tmp/prune/log.txt", "a")
for fn in files:
    path = os.path.join(CSRC, fn)
    idx = 0
    while True:
        src = open(path).read()
        cands = [m for m in PAT.finditer(src) if not re.search(r"GPBC_INLINE\s+\w+\s+$", src[max(0, m.start() - 40):m.start()])]
        if idx >= len(cands): break
        m = cands[idx]
        line_no = src.count("\n", 0, m.start()) + 1
        open(path, "w").write(src[:m.start()] + m.group(1) + "_nonorm(" + src[m.end():])
        t0 = time.time()
        ok = run_tests()
        msg = "%s:%d %s -> %s (%.0fs)" % (fn, line_no, src[m.start():m.start() + 70].split("\n")[0], "REMOVED" if ok else "needed", time.time() - t0)
        print(msg, flush=True); log.write(msg + "\n"); log.flush()
        if not ok:
            open(path, "w").write(src)
            idx += 1
print("done")
