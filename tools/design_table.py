#!/usr/bin/env python3
"""Prints the front-page table of DESIGN.md section 5 from a bench.py line: tools/design_table.py <bench.json>"""
import json, sys
line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rows = [dict(name=line["config"]["name"], value=line["value"], ms=line["ms_per_step"], roofline=line["roofline"], **{k: line["config"].get(k) for k in ("rays_per_sample", "nodes_per_ray", "tests_per_ray")})]
rows += line.get("extra_configs", [])
print("| workload | ms / step | Msamples/s | HBM-side bytes per frame (PMC) = GB/s = frac of 8 TB/s | the bound that applies, and its fraction | counters |")
print("|---|---|---|---|---|---|")
for r in rows:
    ro = r["roofline"]
    hbm = f"{ro['achieved'] * r.get('ms', r.get('ms_per_step')) / 1e3:.2f} GB = {ro['achieved']:.0f} GB/s = **{ro['frac']:.3f}**" if ro.get("traffic") is not None else f"(no current profile) {ro['achieved']:.0f} GB/s model"
    valu = ro.get("valu", {})
    bound = []
    if valu: bound.append(f"VALU issue {valu['issue_frac_guide_2clk']:.2f} of one wave-instruction per 2 clocks per SIMD")
    if ro.get("valu_lane_utilisation") is not None: bound.append(f"lane utilisation {ro['valu_lane_utilisation']:.2f}")
    if ro.get("wait_frac") is not None: bound.append(f"{ro['wait_frac']:.2f} of wave cycles waiting")
    if ro.get("tcc_hit_rate") is not None: bound.append(f"L2 hit {ro['tcc_hit_rate']:.2f}")
    if ro.get("gather"): bound.append(f"dependent record fetches {ro['gather']['frac']:.2f} of the chip's measured gather rate")
    extra = f"; {r['nodes_per_ray']} nodes + {r['tests_per_ray']} tests per ray" if r.get("nodes_per_ray") is not None else ""
    if r.get("walk_nodes_per_ray") is not None:                       # the timed walk is not the reference's own: both sets of counters
        extra = (f"; {r['walk']}: {r['walk_nodes_per_ray']} nodes + {r['walk_tests_per_ray']} tests per ray (the reference's walk: {r['nodes_per_ray']} + {r['tests_per_ray']})"
                 + (f"; {100 * r['cert_chain_per_hit']:.2f} % of the hits through the ancestor chain, {r['cert_fallback_per_hit']:.1e} through the reference's walk" if r["walk"] == "certified" else ""))
    src = ro.get("achieved_source", "").split("(")[-1].split(",")[0] if "PMC" in ro.get("achieved_source", "") else "-"
    print(f"| {r['name']} | {r.get('ms', r.get('ms_per_step'))} | **{r['value']:.0f}** | {hbm} | {'; '.join(bound)}{extra} | `{src}` |")
if "cpu_baseline" in line:
    c = line["cpu_baseline"]
    print(f"| CPU oracle ({c['cores']} threads, kind {c['kind']}) | | {c['value']} | | {c['sample']} | |")
for k in ("value_pipelined_batch", "value_incl_d2h"):
    if k in line: print(f"| {line['config']['name']} {k} | | {line[k]:.0f} | | | |")
for r in rows:
    if "certified_vs_reference_tree" in r:
        d = r["certified_vs_reference_tree"]
        print(f"\n{r['name']}: certified walk vs the reference's own tree, one frame from the same RNG state: {d['pixels_differ']} of {d['pixels']} pixels differ, max abs {d['max_abs']:.3e}")
    if "fast_tree_vs_exact" in r:
        d = r["fast_tree_vs_exact"]
        print(f"\n{r['name']}: fast tree vs exact walk, one frame from the same RNG state: {d['pixels_differ']} of {d['pixels']} pixels differ, max abs {d['max_abs']:.3e}, RMSE {d['rmse']:.3e}")
