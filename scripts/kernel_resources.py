"""List registers / LDS / scratch of the library's kernels whose (mangled) name contains a pattern: python scripts/kernel_resources.py wreg3"""
import re, subprocess, sys, tempfile, shutil, os
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "givepose_amd", os.environ.get("GP_LIB", "libgivepose_hip.so"))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
d = tempfile.mkdtemp()
shutil.copy(lib, d)
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", os.path.join(d, os.path.basename(lib))], capture_output=True, text=True)
for f in sorted(os.listdir(d)):
    if "gfx950" not in f:
        continue
    notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(d, f)], text=True)
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if pat in name:
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            print(name[:110], "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "lds", g("group_segment_fixed_size"), "scratch", g("private_segment_fixed_size"),
                  "spill v/s", g("vgpr_spill_count"), g("sgpr_spill_count"))
shutil.rmtree(d)
