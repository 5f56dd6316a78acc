#!/usr/bin/env python3
"""Write profiles/README.md from the files under profiles/ (kernel-stat CSVs, PMC summaries, bench lines): every number in
the per-round tables is computed here, none is typed by hand (the hand-kept README was one refresh behind).
    python tools/make_profiles_readme.py"""
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        short = re.split(r"[(<]", name)[0]
        key = short + (re.search(r"<[^>]*>", name).group(0) if "<" in name.split("(")[0] else "")
        out.setdefault(key, []).append((int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["Percentage"])))
    return out


def k(st, name):
    v = st.get(name)
    if not v:                                  # template arguments changed between rounds: all instantiations with that prefix together
        v = [x[0] for key, x in sorted(st.items()) if key.startswith(name)]
    if not v:
        return "n/a"
    c = sum(x[0] for x in v)
    avg = sum(x[0] * x[1] for x in v) / c
    mn = min(x[2] for x in v)
    more = f", {len(v)} instantiations" if len(v) > 1 else ""
    return f"avg {avg:.1f} µs over {c} launches (min {mn:.1f}{more})"


def round_section(tag):
    L = [f"## {tag}", ""]
    bl = os.path.join(P, f"{tag}_bench_line.json")
    if os.path.exists(bl):
        d = json.load(open(bl))
        rf, ex = d["roofline"], d.get("extra", {})
        L += [f"* `{tag}_bench_line.json` (`python bench.py`): **{d['ms_per_step']} ms/step = {d['value']} {d['unit']}**, "
              f"`{rf['kernel']}` {rf['kernel_ms']} ms by HIP events -> {rf['achieved']} GB/s = {rf['frac']} of {rf['peak']}; "
              f"traffic {rf['traffic']} B vs algorithmic {rf['algorithmic_bytes_per_launch']} B; "
              f"verified bit-exact vs oracle: {d['config']['verified_bit_exact_vs_oracle']}; cpu_baseline {d['cpu_baseline']['value']} "
              f"{d['cpu_baseline']['unit']} on {d['cpu_baseline']['cores']} threads."]
        if "bvh_build" in ex and "roofline" in ex["bvh_build"]:
            b = ex["bvh_build"]
            L += [f"* BLAS {b['n_tris']} tris: {b['ms']} ms = {b['value']} Mprims/s, bit-exact vs oracle on the timed mesh: {b.get('bit_exact_vs_oracle')}; "
                  f"phases (ms) {b.get('phases_ms')}; depth {b['roofline']['depth']}, sum of active prims {b['roofline']['sum_active_prims']}; "
                  f"770 B/prim/level definition: {b['roofline']['emulating_770B_per_prim_level']['achieved']} GB/s "
                  f"({b['roofline']['emulating_770B_per_prim_level']['frac']}), 44 B: {b['roofline']['binned_44B_per_prim_level']['achieved']} GB/s "
                  f"({b['roofline']['binned_44B_per_prim_level']['frac']}); CPU oracle {b.get('cpu_baseline', {}).get('value')} Mprims/s (1 core)."]
        if "tlas" in ex:
            t, w = ex["tlas"], ex.get("tlas_wide_64k", {})
            L += [f"* TLAS 32768: build {t['build_ms']} ms (bit-exact vs oracle: {t.get('bit_exact_vs_oracle')}; oracle {t.get('cpu_baseline', {}).get('value')} ms), "
                  f"refit {t['refit_queued_ms']} ms queued; 65536 wide: build {w.get('build_ms')} ms, refit {w.get('refit_gpu_ms')} ms, "
                  f"refit after motion bit-exact: {w.get('refit_after_motion_bit_exact_vs_oracle')}."]
        if "trace" in ex:
            t = ex["trace"]
            L += [f"* trace (2000 x 131k-tri stress scene): {t['closest_hit_Mrays_per_s']} Mrays/s closest hit, {t['occlusion_Mrays_per_s']} occlusion; "
                  f"CPU oracle {t.get('cpu_baseline', {}).get('value')} Mrays/s on {t.get('cpu_baseline', {}).get('cores')} threads; harness scene "
                  f"{ex.get('trace_harness_scene', {}).get('closest_hit_Mrays_per_s')} Mrays/s."]
    b2 = os.path.join(P, f"{tag}_bench_line_2ranks_1gpu_gloo.json")
    if os.path.exists(b2):
        d = json.load(open(b2))
        L += [f"* `{tag}_bench_line_2ranks_1gpu_gloo.json` (`VOIDIN_DIST_BACKEND=gloo python bench.py --gpus 2 ...`, two ranks sharing the one GPU: "
              f"functional evidence of the launcher, NOT a scaling number - gloo moves the masks through the host): n_gpus {d['n_gpus']}, scaling {d['scaling']}, "
              f"whole list verified vs oracle: {d['config']['verified_bit_exact_vs_oracle']}, CRC {d['config']['draw_list_crc32']}."]
    b3 = os.path.join(P, f"{tag}_bench_line_torchrun_2ranks_shard.json")
    if os.path.exists(b3):
        d = json.load(open(b3))
        L += [f"* `{tag}_bench_line_torchrun_2ranks_shard.json` (the driver's multi-GPU form, `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 "
              f"--master-addr 127.0.0.1 ... bench.py --gpus 2 --gather shard`, again two ranks on the one GPU over gloo): n_gpus {d['n_gpus']}, "
              f"{d['ms_per_step']} ms/step with both ranks' kernels sharing the device, rank 0's shard list verified vs oracle: {d['config']['verified_bit_exact_vs_oracle']}."]
    ks = os.path.join(P, f"{tag}_bench_kernel_stats.csv")
    if os.path.exists(ks):
        st = stats(ks)
        L += [f"* `{tag}_bench_kernel_stats.csv` (`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline "
              f"--no-verify`): `cull_mask_tiled_kernel<unsigned char>` {k(st, 'cull_mask_tiled_kernel<unsigned char>')}; `mask_scan_kernel` {k(st, 'mask_scan_kernel')}; "
              f"`expand_mask_u8_kernel<true, 0>` {k(st, 'expand_mask_u8_kernel<true, 0>')}; `emit_all_u8_kernel` {k(st, 'emit_all_u8_kernel')}; "
              f"`tlas_build_indexed_kernel<VdTlasNode…>` {k(st, 'tlas_build_indexed_kernel<VdTlasNode')}; `blas_small_kernel` {k(st, 'blas_small_kernel')}."]
    kh = os.path.join(P, f"{tag}_bench_noextra_kernel_stats.csv")
    if os.path.exists(kh):
        st = stats(kh)
        L += [f"* `{tag}_bench_noextra_kernel_stats.csv` (the same with `--no-extra`: only the headline's launches, i.e. without the every-mesh-id-changes leg "
              f"that runs the same kernel in its worse regime): `cull_mask_tiled_kernel<unsigned char>` {k(st, 'cull_mask_tiled_kernel<unsigned char>')}; "
              f"`mask_scan_kernel` {k(st, 'mask_scan_kernel')}; `expand_mask_u8_kernel<true, 0>` {k(st, 'expand_mask_u8_kernel<true, 0>')}."]
    pm = os.path.join(P, f"{tag}_cull_pmc.json")
    if os.path.exists(pm):
        d = json.load(open(pm))
        c, e = d.get("cull_mask_tiled_kernel", {}), d.get("expand_mask_u8_kernel", {})
        if c:
            tr = (2 * c["FETCH_SIZE_KB"] + c["WRITE_SIZE_KB"]) * 1024
            L += [f"* `{tag}_cull_pmc.json` (+ `_fetch_size.csv`, `_write_size.csv`; separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of "
                  f"`bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-extra`): `cull_mask_tiled_kernel` FETCH_SIZE {c['FETCH_SIZE_KB']:.0f} KB, "
                  f"WRITE_SIZE {c['WRITE_SIZE_KB']:.0f} KB per launch -> traffic = 2 x FETCH + WRITE = {tr / 1e9:.4f} GB (gfx950: FETCH_SIZE counts 64 B per 128-B "
                  f"request, MI355X_MICROARCH.md HBM section); `expand_mask_u8_kernel` WRITE_SIZE {e.get('WRITE_SIZE_KB', 0):.0f} KB."]
    bv = os.path.join(P, f"{tag}_bvh_kernel_stats.csv")
    if os.path.exists(bv):
        st = stats(bv)
        L += [f"* `{tag}_bvh_kernel_stats.csv` (`rocprofv3 --kernel-trace --stats -- python3 tools/bench_bvh.py --u 2048 --v 2048 --reps 2`; its stdout: "
              f"`{tag}_bench_bvh.log`): `a_apply_kernel` {k(st, 'a_apply_kernel')}, `a_ranks_kernel` {k(st, 'a_ranks_kernel')}, `a_count_kernel` "
              f"{k(st, 'a_count_kernel')}, `a_scan_kernel` {k(st, 'a_scan_kernel')}, `a_bin_kernel` {k(st, 'a_bin_kernel')}, `a_child_kernel` {k(st, 'a_child_kernel')}, "
              f"`blas_mid_kernel` {k(st, 'blas_mid_kernel')}, `blas_small_kernel` {k(st, 'blas_small_kernel')}."]
    bp, bl = os.path.join(P, f"{tag}_bvh_pmc.json"), os.path.join(P, f"{tag}_blas_levels.log")
    if os.path.exists(bp) and os.path.exists(bl):
        # HBM traffic of one 8.4 M-triangle build per kernel (PMC) over that kernel's time in one build (kernel trace): GB/s per kernel
        pm = json.load(open(bp))
        kern = pm.get("kernels", {})
        times = {}
        for line in open(bl):
            m = re.match(r"\s+(?:void )?(\S.*?)\s+calls\s+(\d+) total\s+([0-9.]+) ms", line)
            if m:
                times[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
        rows = []
        for name, v in kern.items():
            key = next((t for t in times if name.startswith(t) or t.startswith(name[:28])), None)
            if key is None:
                continue
            gb = (v["read_MB"] + v["write_MB"]) / 1e3
            ms = times[key][1]
            if ms > 0.2:
                rows.append((gb / ms, f"`{name}` {gb:.1f} GB in {ms:.2f} ms = {gb / ms:.1f} TB/s"))
        if rows:
            tot = pm.get("total", {})
            L += [f"* `{tag}_bvh_pmc.json` + `{tag}_blas_levels.log` (HBM bytes per build by PMC, read = 2 x FETCH_SIZE, write = WRITE_SIZE; kernel time of one build): "
                  + "; ".join(r[1] for r in sorted(rows, reverse=True)) + f"; the whole build {tot.get('sum_MB', 0) / 1e3:.1f} GB."]
    for extra in sorted(glob.glob(os.path.join(P, f"{tag}_*.log")) + glob.glob(os.path.join(P, f"{tag}_*.txt"))):
        base = os.path.basename(extra)
        if base.endswith("bench_bvh.log") or base.split("_", 1)[1] not in NOTES:
            continue
        first = open(extra).readline().strip()
        L += [f"* `{base}`: {NOTES.get(base.split('_', 1)[1], 'log')} - first line: `{first[:160]}`"]
    return L + [""]


NOTES = {
    "pytest_gpu.log": "tail of `python -m pytest tests -q -m gpu --durations=8` on the GPU box",
    "tlas_time.log": "(round 2, `tools/tlas_time.py`, since removed): TLAS build ms at 32768 / 65536 for VD_TLAS_PHASE2 / VD_TLAS_REFRESH variants, bit-exactness vs oracle",
    "tlas_index_profile.log": "`VD_TLAS_PROFILE=1 tools/tlas_time.py`: in-kernel cycle counts of the indexed TLAS build per query stage (4-wave form)",
    "blas_levels.log": "(rounds 2-3, `tools/gpu_prof_gaps.sh`, since removed): per-level span / busy time of phase A of one 8.4 M-triangle build, launch gaps",
    "expand_pipe_experiment.log": "measured and NOT kept: persistent register-prefetch form of the 80 M-instance expansion",
    "blas_item_sweep.log": "phase A item size sweep (`-DVD_ITEM`), 8.4 M triangles",
    "blas_big_tier_experiment.log": "measured and NOT kept: an LDS tier for 2049..8192-prim segments (in-kernel cycles per phase)",
    "ab_trace.log": "`tools/ab_trace.py` on the stress scene: the walk as shipped that round (r01-r03: ray supply, binning, de-indexed leaves, before the loop was restructured; "
                    "r04: exact walk with the fan-out, launches per call 1..4, the opt-in tight top level; tuning-build counters and the 2 ms timeline: lanes per iteration, "
                    "lane-steps / instance entries / TLAS visits per ray, longest chain of dependent steps)",
    "ab_trace_fan_sweep.log": "fan-out parameters before the age gate: grace (iterations after the last draw) 32..512, live-ray threshold 16..64, launches 1..3",
    "ab_trace_fan_age_sweep.log": "fan-out with the age gate (only rays older than 256 / 512 / 1024 wave iterations fan out), 2 and 3 launches, and what the fan-out's kernels cost the tight-top-level walk",
    "trace_relay_experiment.log": "measured and NOT kept: whole rays repacked into full waves over several launches (lanes per iteration 35 -> 50, every launch slower: the call waits for its longest rays)",
    "blas_small_isa.txt": "`tools/blas_small_isa.py`: instructions per section of phase B's small-node path from the ISA (-DVD_ISA_MARKS), priced per batch",
    "ab_trace_loop.log": "`tools/ab_trace.py` after the stepping loop was restructured: VD_OPT_TRACE_YIELD and VD_OPT_TRACE_WAVES sweeps; tuning-build counters "
                         "(lanes per iteration, lane-steps per ray, longest ray, iterations after the last ray was handed out) at 1 M rays and at 64 / 1024 / 16 384 rays alone",
    "probe_gather.log": "`tools/probe_gather.hip`: what a CU pays for 64 divergent 64-byte fetches per wave-step (own 4 x 16 B / quad-cooperative / two lines), 28 waves per CU, "
                        "working sets in L2 / MALL / HBM",
    "trace_l2.txt": "`tools/gpu_prof_trace_l2.sh`: L1 / L2 / fabric counters of the traversal kernels per launch and per ray (rocprofv3 --pmc, one group per pass)",
    "trace_guard_isa.txt": "the toolchain finding behind the explicit `pos < limit` in the ray refill (ISA excerpt)",
    "tlas_nn_table_sim.log": "`tools/tlas_nn_table_sim.py`: hit rates of a nearest-neighbour table in the literal TLAS chain, by refresh interval",
    "blas_item_isa.txt": "the ISA of `a_ranks_kernel` before / after (every load, wait and barrier in program order) and the same-box A/B series of round 4's BLAS work "
                         "(`tools/gpu_blas_ab_multi.sh`): loads of an item in flight together, the segment head through the scalar cache, kernarg preload, DPP reductions, mid tier, a_child",
    "blas_boundary.log": "in-kernel cycle stamps of `a_boundary_kernel` (-DVD_BOUNDARY_PROF) before / after it was spread over all CUs; `a_eval` / mid-tier evaluation staging",
    "blas_two_stream_experiment.log": "measured and NOT kept: a level's rounds as two halves on two streams; two halves of the build with their own level loops",
    "tlas_two_workgroups.log": "VERDICT r3 item 7, measured and not made the default: the indexed TLAS build's speculative query on a second workgroup of the same XCC (205 vs 188 ms) with the counters that say why; mailbox variants",
    "tlas_chain_lds.log": "`tools/tlas_chain_lds_ab.py`: TLAS builds of <= 4096 instances with the chain's slot arrays in LDS against reading them from memory",
    "ab_cull_small.log": "`tools/ab_cull.py --variants 0,1,2,4,8` at 1 k .. 1 M instances: tile size of the fused single-launch cull (the automatic choice is variant 0)",
    "ab_trace_fan_wps.log": "the fan-out's kernels compiled for 4 / 5 / 6 waves per SIMD (spills against occupancy) on the stress scene",
    "ab_trace_sibling_pruning.log": "VERDICT r4 item 6, measured and NOT kept: sibling jobs of the fan-out publish hits early and read the shared bound at every instance entry (bit-identical, 126 vs 128 Mrays/s)",
    "fuzz.log": "fuzz campaign of the round with fresh seeds (round 6: `tools/fuzz_*.py --seed 6xx`; the fixed-seed slices run inside `pytest -m gpu`).  Round 5's file: 7 200 + 2 700 (after the sparse rank tables; up to 250 k triangles) + 9 600 (final tree) random meshes, 900 + 800 random cull scenes (both forms), 600 + 600 random TLAS scenes, 750 + 600 random trace scenes x 6 walks against the oracle, byte for byte: 0 mismatches",
    "blas_sizes.log": "BLAS build time by mesh size, 131 k .. 32.8 M triangles, one mesh per build",
    "stress.log": "the race / repeat stress tools at the round's final tree: every run of the cull, expansion, refit, BLAS build (incl. a 3 000-mesh batch), TLAS build and traversal compared with the first, bit for bit",
    "blas_issue_cost.log": "what bounds the two kernels of a phase-A round: sparse rank tables (kept), pads of 200 scalar / vector instructions and of dependent loads, empty-grid probes (4.9 us to start a round's 32 768 waves), "
                           "item sizes, wave-uniform short forms, per-item records, host-side round constants (kept), phase B beside the last levels - each with its same-box numbers",
    "blas_mid_ab.log": "VERDICT r4 item 2 (ii), measured and NOT kept: the BLAS mid tier at 4096 / 8192 triangles against 2048",
    "blas_small_classes.log": "`tools/blas_small_classes.py` (round 6, VERDICT r5 item 4): phase B's wave-cycles by node size class (<= 32, 33..64, 65..128, 129..256, 257..512) and step (setup, 21 trials, evaluation, final shuffle, children) from s_memtime brackets in the tuning build",
    "pytest_gpu_final.log": "tail of `pytest -m gpu` at round 6's final tree: 281 passed, 3 skipped (the third skip: the external-semaphore round trip, whose import this HIP runtime refuses)",
    "suite_repeat.log": "the gpu suite four times in a row on one box + smoke(): no flaky test",
    "external_semaphore_probe.log": "what this HIP runtime accepts for the HIP -> wgpu ordering: import of a real DRM sync object (refused), stream value waits on ordinary / signal / fd-imported memory, host functions",
    "blas_bin_stream.log": "`a_bin_kernel` streaming in pos0 order instead of gathering through the final arrangement, then with its loads pipelined across items: same-box A/B, kernel stats, `a_boundary` cycle stamps",
}


def main():
    head = open(os.path.join(P, "README.md")).read().split("<!-- generated below -->")[0].rstrip()
    tags = sorted({os.path.basename(f).split("_")[0] for f in glob.glob(os.path.join(P, "r[0-9][0-9]_*"))} )
    body = [head, "", "<!-- generated below -->", "",
            "Everything below this line is written by `tools/make_profiles_readme.py` from the files it names.", ""]
    for t in tags:
        body += round_section(t)
    open(os.path.join(P, "README.md"), "w").write("\n".join(body))
    print("profiles/README.md: rounds", tags)


if __name__ == "__main__":
    main()
