"""Regenerate scripts/README.md: one row per script (first sentence of its docstring / leading comment) with the profiles/ files that profiles/README.md
attributes to it.  python scripts/make_index.py"""
import ast, glob, os, re
root = os.path.dirname(os.path.abspath(__file__))
prof = open(os.path.join(root, "..", "profiles", "README.md")).read()
rows = []
for f in sorted(glob.glob(root + "/*") + glob.glob(root + "/probes/*") + glob.glob(root + "/repro/*")):
    name = os.path.relpath(f, root)
    if os.path.isdir(f) or "__pycache__" in f or name == "README.md" or not f.endswith((".py", ".sh", ".hip")):
        continue
    txt = open(f, errors="ignore").read()
    desc = ""
    if f.endswith(".py"):
        try:
            d = ast.get_docstring(ast.parse(txt))
            desc = " ".join(d.split()) if d else ""
        except SyntaxError:
            pass
    if not desc:
        desc = " ".join(l.lstrip("#/ ").strip() for l in txt.splitlines()[:12] if l.startswith(("#", "//")) and not l.startswith("#!"))
    desc = re.split(r"(?<=[a-z\)])\. ", desc)[0][:230].replace("|", "/")
    outs = [line.split("|")[1].strip() for line in prof.splitlines() if line.startswith("|") and ("scripts/" + name) in line]
    rows.append((name, desc, "; ".join(outs)[:200]))
out = ["# scripts/", "",
       "Build-time and investigation tooling; nothing here is imported by `givepose_amd/` (the product) or needed at run time.  `profiles/README.md` lists every committed",
       "profile with the command that produced it; this index goes the other way: script -> what it does -> the `profiles/` files it produced (as named there).  Regenerate: `python scripts/make_index.py`.", "",
       "**Oracle pinning (run in the build container, never on the GPU box):** `gen_golden.py`, `gen_golden_dcnv3_any.py`, `gen_golden_scale_net.py`, `ref_shim.py` (INTEGRATION.md section 6).",
       "**Round profiles:** `profile_r0N.sh <commit>` (rocprofv3 kernel traces, PMC traffic, MFMA busy) + `bench_all.sh` (the other BASELINE configs), summarised by `trace_summary.py`, `pmc_traffic.py`, `pmc_calib.py`, `mfma_busy.py`.",
       "**Same-box A/B of the whole forward:** `bench_ab_env.sh \"<ENV=...>\" [pairs]`.", "",
       "| script | what it does | profiles/ files (see profiles/README.md) |", "|---|---|---|"]
out += [f"| `{n}` | {d} | {o} |" for n, d, o in rows]
open(os.path.join(root, "README.md"), "w").write("\n".join(out) + "\n")
print(len(rows), "scripts indexed")
