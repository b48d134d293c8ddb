"""Instruction mix of the library's kernels whose mangled name contains a pattern: python scripts/kernel_isa_stats.py wreg3 [lib]"""
import re, subprocess, sys, tempfile, shutil, os, collections
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "givepose_amd", sys.argv[2] if len(sys.argv) > 2 else "libgivepose_hip.so")
pat = sys.argv[1]
d = tempfile.mkdtemp()
shutil.copy(lib, d)
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", os.path.join(d, os.path.basename(lib))], capture_output=True, text=True)
for f in sorted(os.listdir(d)):
    if "gfx950" not in f:
        continue
    dis = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", os.path.join(d, f)], text=True)
    cur, cnt = None, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if cur and pat in cur:
                tot = sum(cnt.values())
                top = ", ".join(f"{k} {v}" for k, v in cnt.most_common(22))
                print(cur[:90], "total", tot, "|", top)
            cur, cnt = m.group(1), collections.Counter()
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if m and cur:
            cnt[m.group(1)] += 1
    if cur and pat in cur:
        print(cur[:90], "total", sum(cnt.values()), "|", ", ".join(f"{k} {v}" for k, v in cnt.most_common(22)))
shutil.rmtree(d)
