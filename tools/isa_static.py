"""Static instruction mix of the headline kernel, physics_kernel<0,false,true,4> (variant 4_0_0_1), compiled with the product flags of csrc/Makefile and
-save-temps: how many of its vector-ALU instructions are PACKED fp32 (v_pk_*: two flops per lane per issue - what the 157.3 TFLOP/s peak assumes)?
Writes profiles/isa_static.json with the hash of the kernel sources; bench.py quotes `valu_packed_share` and the roofline fraction against the UNPACKED
ceiling (78.65 TFLOP/s) from it when the loaded library was built from the same sources.       python tools/isa_static.py      (here, no GPU)"""
import collections, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phase_guided_terrain_traversal_amd import srchash
csrc = os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "csrc")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -amdgpu-load-store-vectorizer=0".split()
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["hipcc"] + flags + ["-DPG_SUBS=4", "-DPG_MODE=0", "-DPG_DR=0", "-DPG_TERRAIN=1", "-save-temps", "-c", os.path.join(csrc, "pgtt_physics_inst.hip"), "-o", os.path.join(td, "p.o")],
                   check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(td) if f.endswith(".s") and "gfx950" in f][0]
    c = collections.Counter()
    for ln in open(os.path.join(td, asm)):
        t = ln.strip()
        if not t or t[0] in ";." or t.endswith(":") or not t.startswith(("v_", "s_", "ds_", "global_", "scratch_", "buffer_")):
            continue
        m = t.split()[0]
        k = ("accvgpr" if m.startswith("v_accvgpr") else "v_pk" if m.startswith("v_pk") else "v_mfma" if m.startswith("v_mfma") else "valu_dpp" if m.startswith("v_") and "dpp" in t
             else "valu" if m.startswith("v_") else "branch" if m.startswith(("s_cbranch", "s_branch")) else "waitcnt" if m.startswith("s_waitcnt") else "nop" if m.startswith("s_nop")
             else "salu" if m.startswith("s_") else "lds" if m.startswith("ds_") else "vmem")
        c[k] += 1
valu = c["v_pk"] + c["valu"] + c["valu_dpp"]
out = {"kernel": "physics_kernel<0,false,true,4> (4_0_0_1)", "csrc_sha256": srchash.source_sha256(), "static_counts": dict(sorted(c.items())), "valu_arith": valu,
       "valu_packed_share": c["v_pk"] / valu, "note": "static counts of the ISA listing (loops counted once); v_accvgpr_* are register moves to / from the AGPR file, not arithmetic"}
json.dump(out, open(os.path.join(ROOT, "profiles", "isa_static.json"), "w"), indent=1)
print(out)
