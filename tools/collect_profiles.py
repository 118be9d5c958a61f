"""Copy the summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked) and refresh
   profiles/hbm_traffic.json (read by bench.py for roofline.traffic).   usage: python tools/collect_profiles.py r02b "note" """
import json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for a, b in (("driver_cmd_bench.json", "driver_cmd_bench.json"), ("bench_level4.json", "level4_bench.json"), ("bench_flat.json", "flat_bench.json"), ("bench_wfc_dr_8192.json", "wfc_dr_8192_bench.json"),
             ("bench_level4_quad.json", "level4_quad_layout_bench.json"), ("kernel_stats.csv", "level4_kernel_stats.csv"),
             ("bench_wfc_dr_8192_quad.json", "wfc_dr_8192_quad_layout_bench.json"), ("bench_level4_8192.json", "level4_8192_bench.json"),
             ("bench_level4_32768.json", "level4_32768_bench.json"), ("bench_flat_16384.json", "flat_16384_bench.json"),
             ("train_rollout_kernel_stats.csv", "train_rollout_kernel_stats.csv"), ("train_ppo_kernel_stats.csv", "train_ppo_kernel_stats.csv"),
             ("train_run.txt", "train_run.txt"), ("kt_rollout_bench.json", "train_rollout_bench.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f"{tag}_{b}"))
hdr = (f"# rocprofv3 PMC summary, {tag}: {note}; 4096 envs on level4, per-launch means; collected by tools/profile_round.sh\n"
       "# separate passes: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU | --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE\n"
       f"#   -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline ; kernel trace ({tag}_level4_kernel_stats.csv): rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (the default 300 + 30 steps)\n"
       "# FETCH_SIZE / WRITE_SIZE in KB as reported; summarised by tools/pmc_summary.py\n")
body = open(os.path.join(src, "pmc_summary.txt")).read()
open(os.path.join(dst, f"{tag}_level4_pmc_summary.txt"), "w").write(hdr + body)
FETCH_CORR, WRITE_CORR = 2.0, 1.0
# gfx950 correction (MI355X_MICROARCH.md "HBM"; calibrated on this library's own access pattern with tools/probes/traffic_calib.hip -
# 4-byte-per-lane coalesced SoA rows, 256 MiB read and written: FETCH_SIZE reports exactly 1/2 of the bytes read, WRITE_SIZE is exact)
NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes (profiles/%s_%s), KB * 1024, FETCH x 2 (gfx950 correction, calibrated with "
        "tools/probes/traffic_calib.hip: 131 082 KB reported for 262 144 KB read, WRITE exact).  L2 <-> fabric requests incl. Infinity-Cache hits: the 8 XCD L2s "
        "are not coherent with each other, so each launch reads its state rows (1x) and, per XCD, the records of the terrain variants its envs stand on "
        "(randomize.domain_randomize hands the per-env variant draws out in ascending order within blocks of 4096 global env ids - the product default since round 4 - so an XCD's range of envs stands on ~1/8 of the 0.8 MB table; in draw order, `--unsorted-variants`, every XCD reads all of it); writes are 16-byte pieces of "
        "64-byte requests per wave (4 envs per wave in the hex layout, 8 in oct).  valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES.")
tp = os.path.join(dst, "hbm_traffic.json")
t = json.load(open(tp))


def entry(text, fname):
    val = {}
    for ln in text.splitlines():
        m = re.match(r"void pgtt::physics_kernel<0.*?\s(\w+)\s+launches=\s*\d+ mean=([\d.e+]+)", ln)
        if m:
            val[m.group(1)] = float(m.group(2))
    e = {"physics_bytes_per_launch": FETCH_CORR * val["FETCH_SIZE"] * 1024 + WRITE_CORR * val["WRITE_SIZE"] * 1024,
         "fetch_bytes": FETCH_CORR * val["FETCH_SIZE"] * 1024, "write_bytes": WRITE_CORR * val["WRITE_SIZE"] * 1024,
         "raw_counters_KB": {"FETCH_SIZE": val["FETCH_SIZE"], "WRITE_SIZE": val["WRITE_SIZE"]}, "note": NOTE % (tag, fname)}
    if "SQ_ACTIVE_INST_VALU" in val and "SQ_WAVE_CYCLES" in val:
        e["valu_busy"] = val["SQ_ACTIVE_INST_VALU"] / val["SQ_WAVE_CYCLES"]
        e["wait_any"] = val.get("SQ_WAIT_ANY", 0.0) / val["SQ_WAVE_CYCLES"]
        e["valu_insts_per_launch"] = val.get("SQ_INSTS_VALU"); e["waves"] = val.get("SQ_WAVES")
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in val:
        e["mfma_ops"] = val["SQ_INSTS_VALU_MFMA_MOPS_F32"]
    return e


t["level4_4096"] = entry(body, "level4_pmc_summary.txt")
for T, key in (("flat", "flat_4096"), ("wfc_dr_8192", "wfc_dr_8192"), ("level4_unsorted", "level4_4096_unsorted_variants")):
    f = os.path.join(src, f"pmc_summary_{T}.txt")
    if os.path.exists(f):
        txt = open(f).read()
        open(os.path.join(dst, f"{tag}_{T}_pmc_summary.txt"), "w").write(hdr.replace("4096 envs on level4", T) + txt)
        t[key] = entry(txt, f"{T}_pmc_summary.txt")
# which kernel sources these counters belong to: bench.py withholds them ("profile_stale": true) when the library it times was built from others
sys.path.insert(0, root)
from phase_guided_terrain_traversal_amd import native
info = native.build_info()
assert info["flavor"] == "product" and info["src"] == native.source_sha256(), "libpgtt.so is not the product build of the sources on disk: rebuild before collecting"
t["_source"] = {"csrc_sha256": info["src"], "flavor": info["flavor"], "lib_sha256": native.library_sha256(), "tag": tag,
                "what": "pgtt_build_info() of the libpgtt.so the counters were collected on: SHA-256 over the sources of physics_kernel (csrc/pgtt_physics_inst.hip and the headers it includes, csrc/Makefile, include/pgtt.h: srchash.py) embedded at build time; lib_sha256 = the file itself (informational)"}
json.dump(t, open(tp, "w"), indent=1)
print("collected", tag, {k: (round(v["physics_bytes_per_launch"] / 1e6, 2), round(v.get("valu_busy", 0), 3)) for k, v in t.items() if not k.startswith("_")})
