"""The GEMM chain's closing table: for each of the eight GEMM calls of the 100,000-patch training step

    achieved us                      HIP events around the call inside the running step (tools/ab_step.py, this box, this run)
    MFMA-only us                     3 x 2MNK fp16-MFMA flops / the rate a register-only v_mfma_f32_32x32x16_f16 loop sustains on random operands
                                     at the socket's power cap (tools/ubench/mfma_power arm 0, SAME box, same run)
    traffic us                       the call's HBM bytes (PMC FETCH_SIZE x2 + WRITE_SIZE of the same step, tools/pmc_step.py) / the rate a pure
                                     streaming kernel with the same read:write mix reaches (tools/ubench/hbm_mix, same box)
    sum, max                         the two ways the floors can combine

and the reading: the socket sits at its 1,400 W cap for the whole step, so time is ENERGY / power: the energy of the matrix work and the
energy of moving the bytes ADD (they draw from one budget; overlapping them in time lowers the clock of both), which is why every call lands
on MFMA-only + traffic and not on their maximum. Usage:
    python tools/gemm_closing_table.py ab.txt mfma_power.txt hbm_mix.txt step_traffic_dir [N] [power_arms.txt]  > profiles/rNN_gemm_closing_table.md

Round 6: the additive form above prices BOTH arms at the full cap, i.e. it counts the board's idle power twice and assumes the
streaming arm draws the cap, which nobody had measured. With power_arms.txt (tools/power_arms.sh: board power of every arm, sampled while it runs
alone) the table gains the corrected floor: only DYNAMIC energy adds, and the budget it draws on is cap - idle:

    floor = [ t_mfma * (P_mfma - P_idle) + t_traffic * (P_stream - P_idle) ] / (P_cap - P_idle)"""
import collections, csv, glob, re, sys

ab, mfma, hbm, tdir = sys.argv[1:5]
N = int(sys.argv[5]) if len(sys.argv) > 5 else 100000
power = {}
if len(sys.argv) > 6 and __import__('os').path.exists(sys.argv[6]):
    for ln in open(sys.argv[6]):
        m = re.match(r"(\w+)\s+samples\s+\d+\s+W min\s+[\d.]+ mean\s+([\d.]+)", ln)
        if m:
            power[m.group(1)] = float(m.group(2))
        m = re.search(r"Max Graphics Package Power \(W\):\s*([\d.]+)", ln)
        if m:
            power["cap"] = float(m.group(1))
GEMMS = [("fwd1", 1024, 512, "NT"), ("fwd2", 512, 512, "NT"), ("fwd_ab", 512, 768, "NT"), ("wgrad_ab", 512, 768, "TN"), ("dgrad_ab", 768, 512, "NT"),
         ("wgrad2", 512, 512, "TN"), ("dgrad2", 512, 512, "NT"), ("wgrad1", 1024, 512, "TN")]
# ---- achieved: mean over the fp32-bag lines of ab_step.py
rows = [ln for ln in open(ab) if "/fp32" in ln and "step" in ln]
ach = collections.defaultdict(list)
step_ms = []
for ln in rows:
    step_ms.append(float(re.search(r"step\s+([\d.]+) ms", ln).group(1)))
    for name, *_ in GEMMS:
        ach[name].append(float(re.search(r"\b%s\s+([\d.]+)" % name, ln).group(1)))
    ach["pool_fwd"].append(float(re.search(r"pool_fwd\s+([\d.]+)", ln).group(1)))
ach = {k: sum(v) / len(v) for k, v in ach.items()}
# Round 6: calls of at most 262,144 rows run their three weight gradients as ONE launch at the end of the backward (gemm_tn_h2_batch_kernel); the
# library books that launch (+ the slab reduction) under the first weight gradient's event pair and leaves the other two pairs empty.
BATCHED = ach["wgrad2"] < 20.0 and ach["wgrad1"] < 20.0
if BATCHED:
    ach["wgrad_x3"] = ach["wgrad_ab"] + ach["wgrad2"] + ach["wgrad1"]
    GEMMS = [g for g in GEMMS if g[3] == "NT"] + [("wgrad_x3", 0, 0, "TN")]
# ---- MFMA-only rate (arm 0, random operands), TFLOP/s of fp16 MFMA
m = re.search(r"arm 0[^:]*:\s*[\d.]+ ms,\s*([\d.]+) TFLOP/s.*?shader clock ([\d-]+) MHz", open(mfma).read())
mfma_tf, mfma_clk = float(m.group(1)), m.group(2)
# ---- streaming rates, TB/s
h = open(hbm).read()
rate = {"read": float(re.search(r"read only \(2 streams\)\s+[\d.]+ ms\s+([\d.]+) TB/s", h).group(1)),
        "copy": float(re.search(r"copy\s+1 R : 1 W\s+[\d.]+ ms\s+([\d.]+) TB/s", h).group(1)),
        "2r1w": float(re.search(r"2 R : 1 W\s+[\d.]+ ms\s+([\d.]+) TB/s", h).group(1))}
# ---- traffic per dispatch of the LAST step (FETCH_SIZE x 2 for gfx950's wide reads, KB units)
def last_step(counter):
    f = glob.glob(f"{tdir}/{counter}/**/*counter_collection.csv", recursive=True)[0]
    seq = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            seq.append((int(r["Dispatch_Id"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("toad::", ""), float(r["Counter_Value"])))
    seq.sort()
    per = collections.OrderedDict()
    for did, name, v in seq:
        per.setdefault(did, [name, 0.0])[1] += v
    disp = list(per.values())
    gem = [d for d in disp if d[0].startswith("gemm_nt_h2_big") or d[0].startswith("gemm_tn_h2_big") or d[0].startswith("gemm_tn_h2_batch")]
    return gem[-(6 if BATCHED else 8):]
fe, wr = last_step("FETCH_SIZE"), last_step("WRITE_SIZE")
order = ["fwd1", "fwd2", "fwd_ab", "wgrad_ab", "dgrad_ab", "wgrad2", "dgrad2", "wgrad1"]     # launch order inside toad_mil_step_f32 (csrc/step.hip)
if BATCHED:
    order = ["fwd1", "fwd2", "fwd_ab", "dgrad_ab", "dgrad2", "wgrad_x3"]
traffic = {}
for name, (kf, f_kb), (kw, w_kb) in zip(order, fe, wr):
    assert kf == kw, (kf, kw)
    traffic[name] = (2 * f_kb * 1024 / 1e6, w_kb * 1024 / 1e6, kf)
print(f"# closing table of the GEMM chain, one {N:,}-patch training step on a raw fp32 bag (same box, same gpurun call for every column)\n")
print(f"* achieved: HIP events around each call inside the running step (`tools/ab_step.py {N} 30`, mean of {len(rows)} runs; step {sum(step_ms) / len(step_ms):.3f} ms)")
print(f"* MFMA-only: `tools/ubench/mfma_power` arm 0 (register operands, random fp16 data) sustains **{mfma_tf:.0f} TFLOP/s** of `v_mfma_f32_32x32x16_f16` at the "
      f"1,400 W cap (shader clock {mfma_clk} MHz; nominal 2,500 at 2.4 GHz) = {mfma_tf / 3:.0f} TF fp32-equivalent for three-term products")
print(f"* traffic: PMC bytes of the same step (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) at the rate `tools/ubench/hbm_mix` reaches for that mix "
      f"(read-only {rate['read']:.2f}, 1R:1W {rate['copy']:.2f}, 2R:1W {rate['2r1w']:.2f} TB/s)\n")
have_p = all(k in power for k in ("idle", "mfma_only", "copy_random", "r2w1_random", "cap"))
if have_p:
    dyn_cap = power["cap"] - power["idle"]
    w_m = (power["mfma_only"] - power["idle"]) / dyn_cap
    w_t = {"copy": (power["copy_random"] - power["idle"]) / dyn_cap, "2r1w": (power["r2w1_random"] - power["idle"]) / dyn_cap}
    print(f"* board power of each arm running alone (`tools/power_arms.sh`, rocm-smi, same box, same call): idle **{power['idle']:.0f} W**, MFMA-only loop "
          f"**{power['mfma_only']:.0f} W**, streaming 1R:1W on random data **{power['copy_random']:.0f} W** (zeros: {power.get('copy_zeros', float('nan')):.0f}), 2R:1W "
          f"**{power['r2w1_random']:.0f} W** (zeros: {power.get('r2w1_zeros', float('nan')):.0f}), read-only {power.get('read2_random', float('nan')):.0f} W; cap {power['cap']:.0f} W. "
          f"Dynamic-energy weights: MFMA {w_m:.2f}, 1R:1W {w_t['copy']:.2f}, 2R:1W {w_t['2r1w']:.2f} of the cap's dynamic budget ({dyn_cap:.0f} W)\n")
print("| call | shape (M x K -> N) | GFLOP (2MNK) | achieved us | TF-eq | MFMA-only us | read MB | written MB | traffic us | MFMA-only + traffic | achieved / (sum) | achieved / max |"
      + (" dynamic-energy floor us | achieved / floor |" if have_p else ""))
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|" + ("---:|---:|" if have_p else ""))
tot = collections.Counter()
for name, k, n, kind in GEMMS:
    flop = 2.0 * N * k * n if name != "wgrad_x3" else 2.0 * N * (512 * 768 + 512 * 512 + 1024 * 512)
    t_m = 3 * flop / (mfma_tf * 1e12) * 1e6
    rd, wrb, kern = traffic[name]
    mix = "2r1w" if kind == "TN" else "copy"                        # TN: two activation streams in, slabs out; NT: one stream in, one out
    t_t = (rd + wrb) * 1e6 / (rate[mix] * 1e12) * 1e6
    a = ach[name]
    fl = (t_m * w_m + t_t * w_t[mix]) if have_p else 0.0
    shape = f"{N:,} x {k} -> {n}" if name != "wgrad_x3" else f"three products over {N:,} rows, one launch + slab reduction"
    print(f"| {name} (`{kern.split('<')[0]}`) | {shape} | {flop / 1e9:.1f} | {a:.1f} | {flop / a / 1e6:.0f} | {t_m:.1f} | {rd:.0f} | {wrb:.0f} | {t_t:.1f} | "
          f"{t_m + t_t:.1f} | {a / (t_m + t_t):.2f} | {a / max(t_m, t_t):.2f} |" + (f" {fl:.1f} | {a / fl:.2f} |" if have_p else ""))
    tot["a"] += a; tot["m"] += t_m; tot["t"] += t_t; tot["f"] += flop; tot["fl"] += fl
print(f"| **chain** | | {tot['f'] / 1e9:.0f} | **{tot['a']:.0f}** | {tot['f'] / tot['a'] / 1e6:.0f} | {tot['m']:.0f} | | | {tot['t']:.0f} | {tot['m'] + tot['t']:.0f} | "
      f"**{tot['a'] / (tot['m'] + tot['t']):.2f}** | {tot['a'] / max(tot['m'], tot['t']):.2f} |" + (f" {tot['fl']:.0f} | **{tot['a'] / tot['fl']:.2f}** |" if have_p else ""))
print(f"\n`roofline_mfma.frac` against the nominal 833.3 TF: {tot['f'] / tot['a'] / 1e6 / 833.3:.3f}; against the power-capped MFMA-only rate ({mfma_tf / 3:.0f}): "
      f"{tot['f'] / tot['a'] / 1e6 / (mfma_tf / 3):.3f}; against MFMA-only + traffic (energy-additive floor): {(tot['m'] + tot['t']) / tot['a']:.3f}.")
if have_p:
    print(f"\nCorrected reading (round 6): with the idle power subtracted and every arm priced at the power it was MEASURED to draw, the chain's floor is "
          f"{tot['fl']:.0f} us and the chain runs at **{tot['a'] / tot['fl']:.2f}** of it ({(1 - tot['fl'] / tot['a']) * 100:.0f} % above). The round-5 form (both arms at the full cap) "
          f"gave {tot['m'] + tot['t']:.0f} us / {tot['a'] / (tot['m'] + tot['t']):.2f}: it double-counted {power['idle']:.0f} W of idle power and assumed a streaming kernel draws the cap.")
