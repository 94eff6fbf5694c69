#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): forward samples/sec of the DLRM sparse stage on synthetic
Criteo-shaped batches — 65 536 samples x 26 sparse fields x dim 128, 1M rows per table — on N
MI355X GPUs of one node.  `--workload` selects the other BASELINE configs for the same contract.

  --workload dlrm_fused (default, configs[1]): ONE fused gather + pairwise-dot launch per step
      (rec_gather_pairwise_dot_f32: ids -> 26 embedding rows + the bottom-MLP vector -> 351 dots ++ dense).
  --workload gather : the materialised gather+concat (rec_gather_concat_f32) — the north-star roofline kernel;
      also measured beside the headline as "gather_roofline".
  --workload autoint (configs[2]) : AutoInt 39 fields x dim 16, 3 interacting layers, 2 heads, batch 4096.
  --workload din     (configs[3]) : DIN history lookup + attention pooling, T <= 100, d = 192, batch 8192.
  --workload sasrec  (configs[4]) : SASRec S = 200, d = 64, 1 block, 100 negatives, batch 8192 (global), forward.

A step = one pass of the hot path over one batch already resident in HBM.  Steps rotate over 8 distinct
id batches, so no launch re-reads the rows of the launch before it out of the 256-MiB Infinity Cache.
Untimed: a device spin-up (>= 0.3 s of steps: clocks and TLBs settle; the first ~100 ms after idle run up to
25 % slower) and the W warm-up steps.  Timed: EXACTLY K steps between barrier + synchronize on both sides;
`value` = units of all ranks / max-over-ranks wall time.  A second, separate pass brackets every launch with
its own HIP events for the p10/p50/p90 in `launch_us`.

The default N = 1 line also carries, measured in the same run: `gather_roofline` (the north-star kernel), `zipf` (the
headline step on Zipf(1.05) ids), `configs` (BASELINE configs[2..4]: autoint / din / sasrec with their own rooflines),
`placed` (see below), `pcie_inclusive`, `cpu_baseline`.

Table placement: the time of both DLRM kernels depends on which physical memory the 13.3 GB of tables received (same
box, same kernel: gather 300-329 us, fused 165-182 us by allocation).  Round 3 looked for the cause (profiles/
r03_placement_probe.txt: not the allocation call — hipMalloc, contiguous, VMM chunks of 2 MiB..1 GiB all show the same
per-arena spread; the per-arena UTCL1 / multi-miss counters differ, i.e. the page-table fragments the driver builds for
whatever physical blocks it had) and found no user-space control.  The HEADLINE therefore runs on ONE PLAIN ALLOCATION
(--arena-candidates 1, the default); the measured placement (recamd.ops.place_table_arena: N arenas side by side, this
workload's step timed on each, the fastest kept) is reported beside it as `placed` (N = --placed-candidates, 0 = skip).

Multi-GPU: `python bench.py --gpus N` starts its own N rank processes (one per GPU, before anything touches a GPU) when
no launcher did (no WORLD_SIZE in the environment); under `python -m torch.distributed.run` it is one of the ranks.
Weak scaling, the same batch per GPU.
  --placement replicated (default): every GPU holds all tables (13.3 GB of 288 GB), as the reference's
      MirroredStrategy mirrors its variables (src/ctr/fm/train.py:43); the forward has no collective.
  --placement rowshard: tables row-sharded cyclically (owner = id % G) behind the RCCL all-to-all pair
      (recamd.dist.ShardedTables): rows this rank owns are read in place, ids + rows of batch i+1 travel on a
      communication stream while batch i's fused kernel runs — xGMI-bound for uniform ids.
  With N > 1 the replicated line also carries a "rowshard" object measured in the same run (and vice versa
  the placement is named in config.placement), so a scaling run reports both; `exchange` says which transport ran
  and how many ranks the RCCL communicator itself reported.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "recommend-tf2.0_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BF16_DENSE_PEAK_TF = 2500.0    # dense bf16 MFMA peak; a bf16x3 product costs 6 MFMAs -> 416.7 TF fp32-equivalent
F32_MFMA_PEAK_TF = 157.3       # dense fp32 MFMA peak (= the fp32 vector peak)
NB = 8                         # distinct id batches rotated over the steps


def pmc_traffic(kernel_substr, cfg):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic[_<workload>].json:
    FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc runs of this same bench).  PMC counters cannot be
    read from inside the timed run, so `traffic` is the latest committed measurement for the SAME workload shape, or None."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")))):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        c = d.get("config", {})
        if any(c.get(k) != v for k, v in cfg.items()):
            continue
        for k, v in d.get("kernels", {}).items():
            if kernel_substr in k:
                return int(v["traffic_bytes_per_launch"]), os.path.basename(f)
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["dlrm_fused", "gather", "autoint", "din", "sasrec"], default="dlrm_fused")
    ap.add_argument("--placement", choices=["replicated", "rowshard"], default="replicated")
    ap.add_argument("--ids", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (0 = the BASELINE config's batch)")
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--vocab", type=int, default=0, help="rows per table (0 = the BASELINE config's vocabulary)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--arena-candidates", type=int, default=1,
                    help="table arenas allocated and probed before one is kept for the HEADLINE (1 = a plain allocation)")
    ap.add_argument("--arena-skip-gb", type=float, default=0.0,
                    help="plain allocation: GiB of device memory held by a spacer while the tables are allocated, then freed "
                         "(the tables then do not share the low end of a fresh process's memory with the runtime's own allocations)")
    ap.add_argument("--placed-candidates", type=int, default=6,
                    help="side measurement `placed`: the headline step on the best of this many arenas (0 = skip)")
    ap.add_argument("--din-width", type=int, default=64, help="--workload din: table width (3 tables; BASELINE configs[3] = 64)")
    ap.add_argument("--transport", choices=["cabi", "torch"], default=None,
                    help="rowshard: 'cabi' = the library's RCCL communicator, 'torch' = torch.distributed collectives "
                         "(default: cabi under the nccl backend, with a visible fallback to torch)")
    ap.add_argument("--cache-rows", type=int, default=0, help="rowshard: hot-row replicas per rank (0 = off)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-spawned N > 1 run: watchdog of the rank processes (s)")
    ap.add_argument("--spinup", type=float, default=0.3, help="seconds of untimed steps before the warm-up")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--cpu-samples", type=int, default=16384)
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (gather_roofline, rowshard)")
    ap.add_argument("--side-timeout", type=float, default=150.0, help="watchdog of the N>1 side placement measurement (s)")
    ap.add_argument("--force", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B only: rec_debug_force(KEY, VALUE) before the workload is built (kernel variants a shape would not select)")
    ap.add_argument("--side-leg", action="store_true", help=argparse.SUPPRESS)   # internal: child process of an N>1 run
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------
# workloads: each returns a dict with  step(i), units (samples per step), bytes/flops the dominant kernel needs,
# the roofline labels and the config description
# --------------------------------------------------------------------------------------------------
def make_ids(torch, dev, a, B, F, V, rank, shape=None):
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    out = []
    for j in range(NB):
        if a.ids == "uniform":
            out.append(torch.randint(0, V, shape or (B, F), device=dev, dtype=torch.int32, generator=gen))
        else:  # Zipf(1.05) clipped to V — Criteo-like skew
            import numpy as np
            z = np.random.default_rng(1 + rank + 1000 * j).zipf(1.05, size=shape or (B, F))
            out.append(torch.from_numpy(((z - 1) % V).astype("int32")).to(dev))
    return out


def wl_dlrm(torch, dev, a, rank, world, fused=True):
    from recamd import ops
    B, F, V, D = a.batch or 65536, a.fields, a.vocab or 1_000_000, a.dim
    n = F + 1
    P = n * (n - 1) // 2
    gen = torch.Generator(device=dev).manual_seed(0)
    sharded = None
    placed = None
    # a plain allocation (the headline): the tables are the FIRST allocation of the process.  With placement by
    # measurement the ids, dense and result buffers come first instead: buffers that land in memory recycled from the
    # freed arena candidates ran the gather 2 % slower (tools/exp/arena_drift.py)
    early_arena = None
    if a.placement == "replicated" and a.arena_candidates <= 1 and os.environ.get("REC_BENCH_TABLES_LAST") != "1":
        spacer = torch.empty(int(a.arena_skip_gb * 2 ** 30), dtype=torch.uint8, device=dev) if a.arena_skip_gb > 0 else None
        early_arena = torch.empty((F, V, D), dtype=torch.float32, device=dev)
        del spacer
        torch.cuda.empty_cache()
    ids = make_ids(torch, dev, a, B, F, V, rank)
    dense = torch.rand((B, D), device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    # (B, 479) result with a 480-float (16-B aligned) row stride, as recamd.ops allocates it
    out_fused = torch.empty((B, (P + D + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :P + D]
    out_gather = torch.empty((B, F * D), dtype=torch.float32, device=dev)
    if a.placement == "replicated":
        # the arena is placed by measurement (recamd.ops.place_table_arena): which physical memory an allocation gets
        # changes the random-row read rate by up to 7 % on one box; every candidate's probe time is reported
        p_fused = lambda g, i: ops.gather_pairwise_dot(g, ids[i % NB], dense, out=out_fused)  # noqa: E731
        p_gather = lambda g, i: ops.gather_concat(g, ids[i % NB], out=out_gather)  # noqa: E731
        # the headline run also reports the materialised gather on the same tables: both kernels are probed (they rank
        # allocations differently) and the sum of their normalised times decides
        arena, placed = ops.place_table_arena(F, V, D, dev, candidates=a.arena_candidates, first=early_arena,
                                              probe=[p_fused, p_gather] if fused else p_gather,
                                              probe_name="[rec_gather_pairwise_dot_f32, rec_gather_concat_f32] steps of this "
                                                         "workload, normalised times added" if fused
                                              else "this workload's step (rec_gather_concat_f32)")
        arena.uniform_(-0.05, 0.05, generator=gen)  # keras 'random_uniform' (dlrm/model.py:34)
        group = ops.TableGroup([arena[f] for f in range(F)])
    else:
        from recamd.dist import ShardedTables
        # shard + replica cache + receive slots in ONE allocation (the row space the consumer kernels address)
        sharded = ShardedTables.empty(F, [V] * F, D, rank, world, dev, max_ids=B * F, cache_rows=a.cache_rows,
                                      cache_refresh_every=16 if a.cache_rows else 0, transport=a.transport)
        arena = sharded.arena
        arena.uniform_(-0.05, 0.05, generator=gen)
        group = None

    def pipelined(consume):
        # plan two batches ahead, ids + rows one batch ahead (communication stream), consume this one (compute stream):
        # the exchange of batch i+1 overlaps batch i's kernel, and the plan's count matrix is on the host before the
        # exchange that needs it is issued.  The pipeline runs on its OWN batch counter: bench passes (spin-up, warm-up,
        # timed, per-launch) each restart `i` at 0, the lookups in flight carry over from one pass to the next.
        pos = [0]

        def step(_i):
            i = pos[0]
            pos[0] += 1
            sharded.prefetch(ids[(i + 2) % NB])
            sharded.prefetch(ids[(i + 1) % NB], rows=True)
            consume(ids[i % NB])
        return step

    if sharded is None:
        step_fused = lambda i: ops.gather_pairwise_dot(group, ids[i % NB], dense, out=out_fused)  # noqa: E731
        step_gather = lambda i: ops.gather_concat(group, ids[i % NB], out=out_gather)  # noqa: E731
    else:
        step_fused = pipelined(lambda t: sharded.lookup_pairwise_dot(t, dense, out=out_fused))
        step_gather = pipelined(lambda t: sharded.lookup(t, out=out_gather))

    bytes_fused = B * (F * D * 4 + F * 4 + D * 4 + (P + D) * 4)     # 15 844 B/sample at 26x128
    bytes_gather = B * F * (2 * D * 4 + 4)                            # 26 728 B/sample at 26x128
    shape_cfg = {"batch": B, "fields": F, "vocab": V, "dim": D, "ids": a.ids}
    w = {
        "units": B, "dtype": "f32", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "shape_cfg": shape_cfg,
        "config": {"batch_per_gpu": B, "global_batch": B * world, "fields": F, "vocab_per_table": V, "dim": D,
                   "ids": a.ids, "id_batches_rotated": NB},
        "arena": arena, "ids": ids, "dense": dense, "sharded": sharded, "out_fused": out_fused, "group": group,
    }
    if placed is not None:
        w["config"]["table_placement"] = placed
    if fused:
        w.update(step=step_fused, work=bytes_fused,
                 kernel=(("rec::pairdot_ring_kernel<27, true, true, 2, 4, 0, 15>" if (F, D) == (26, 128) else
                          "rec::pairdot_ring_gen_kernel (D = %d, n = %d)" % (D, n)) +
                         " (LDS-DMA ring + fp32 MFMA, write-through result stores: fused gather + pairwise dot)")
                 if sharded is None else "row-sharded lookup (RCCL all-to-all pair) + pairwise dot",
                 pmc_key="pairdot_ring_kernel",
                 workload="DLRM %d sparse x %s vocab x dim %d, batch %d/GPU: fused embedding gather + pairwise-dot%s"
                          % (F, "1M" if V == 1_000_000 else str(V), D, B,
                             " (BASELINE configs[1])" if (B, F, V, D) == (65536, 26, 1_000_000, 128) else ""),
                 side_gather=(step_gather, bytes_gather) if sharded is None else None)
    else:
        w.update(step=step_gather, work=bytes_gather, kernel="rec::gather_uniform_kernel<32, 0>",
                 pmc_key="gather_uniform_kernel",
                 workload="DLRM-shape materialised embedding gather+concat, batch 65536/GPU", side_gather=None)
    return w


def wl_autoint(torch, dev, a, rank, world):
    from ctr.autoint.model import AutoInt
    B, F, nd, D, V = a.batch or 4096, 26, 13, 16, a.vocab or 100_000
    fc = [[{'feat': f'I{i}'} for i in range(nd)], [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': D} for i in range(F)]]
    m = AutoInt(fc, att_hidden_units=16, head_num=2, att_layer_num=3, use_res=True)
    dense = torch.rand((B, nd), device=dev)
    ids = make_ids(torch, dev, a, B, F, V, rank)
    flop = 1_382_784  # SURVEY §8d per sample (QKV + QK^T + PV + residual projections of the 3 layers)

    def step(i):
        m([dense, ids[i % NB]])

    return {"step": step, "units": B, "work": B * flop, "dtype": "f32",
            "bound": "mfma", "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "kernel": "rec::mha_ctr_stack_kernel<3, 1, 2, relu, fused io> (rec_autoint_forward_f32: lookup + dense-field "
                      "embedding + 3 interacting layers on fp32 MFMA + Dense(1) + sigmoid in ONE launch)", "pmc_key": None,
            "workload": "AutoInt 39 fields dim 16, 3-layer 2-head self-attn, batch 4096 (BASELINE configs[2])",
            "config": {"batch_per_gpu": B, "global_batch": B * world, "fields": F + nd, "dim": D, "layers": 3, "heads": 2,
                       "flop_per_sample": flop, "peak_note": "157.3 TFLOP/s dense fp32 MFMA (v_mfma_f32_16x16x4_f32)"},
            "side_gather": None, "shape_cfg": {}}


def wl_din(torch, dev, a, rank, world):
    from recamd import ops
    B, T, V = a.batch or 8192, 100, a.vocab or 1_000_000
    Dt = a.din_width
    ntab = 3 if Dt <= 64 else 2
    d = ntab * Dt
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    tabs = [torch.empty((V, Dt), device=dev).uniform_(-0.05, 0.05, generator=gen) for _ in range(ntab)]
    g = ops.TableGroup(tabs)
    ids, real_slots = [], 0
    for j in range(NB):
        lens = torch.randint(1, T + 1, (B,), device=dev, generator=gen)
        x = torch.randint(1, V, (B, T, ntab), device=dev, dtype=torch.int32, generator=gen)
        x[torch.arange(T, device=dev)[None, :] < (T - lens)[:, None]] = 0   # pre-padding with id 0 (pad_sequences)
        ids.append(x)
        real_slots += int(lens.sum().item())
    q = torch.rand((B, d), device=dev, generator=gen)
    W = torch.randn((4 * d, 1), device=dev, generator=gen) * 0.05
    b = torch.zeros(1, device=dev)
    out = torch.empty((B, d), device=dev)

    def step(i):
        ops.gather_din_attention_pool(q, g, ids[i % NB], None, W, b, 'sigmoid', mask_from_ids=True, out=out)

    # bytes the kernel needs: ids of all T slots, rows of the REAL slots only (padded slots carry softmax weight
    # exactly 0 and are not fetched), q in, pooled row out
    need = B * (T * ntab * 4 + 2 * d * 4) + (real_slots / NB) * d * 4
    return {"step": step, "units": B, "work": need, "dtype": "f32", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "kernel": "rec::din_gather_pool_grp_kernel<0, %d, %d> (fused history lookup + attention pooling, one %d-lane group per slot)" % (ntab, Dt // 4, Dt // 4),
            "pmc_key": "din_gather_pool_grp_kernel",
            "workload": "DIN var-len user history (max 100) attention pooling, batch 8192" +
                        (" (BASELINE configs[3])" if (Dt, B, T) == (64, 8192, 100) else ", %d tables x width %d" % (ntab, Dt)),
            "config": {"batch_per_gpu": B, "global_batch": B * world, "maxlen": T, "d": d, "tables": ntab, "width": Dt, "vocab_per_table": V,
                       "mean_real_slots": round(real_slots / NB / B, 2),
                       "bytes_note": "rows of real (non-pad) history slots only + all ids + q + out"},
            "side_gather": None, "shape_cfg": {"workload": "din", "batch": B, "vocab": V, "ids": a.ids, "width": Dt}}


def wl_sasrec(torch, dev, a, rank, world):
    from match.sasrec.model import SASRec
    Bg = a.batch * world if a.batch else 8192
    B = max(1, Bg // world)                       # config 5: global batch 8192, 1024 per GPU at 8 GPUs
    S, n, V, d = 200, 100, a.vocab or 10_000_000, 64
    uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': d},
          {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': d},
          {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': d}]
    kw = {}
    if a.placement == "rowshard":
        from recamd.dist import ShardedTables
        kw = {"sharded": (rank, world),
              "shard_factory": lambda tabs, vocabs, r, w: ShardedTables(tabs, vocabs, r, w, transport=a.transport)}
    m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n, **kw)
    gen = torch.Generator(device=dev).manual_seed(5 + rank)
    batches, real = [], 0
    for j in range(NB):
        lens = torch.randint(1, S + 1, (B,), device=dev, generator=gen)
        seq = torch.randint(1, V, (B, S), device=dev, dtype=torch.int32, generator=gen)
        seq[torch.arange(S, device=dev)[None, :] < (S - lens)[:, None]] = 0
        pos = torch.randint(1, V, (B, 1), device=dev, dtype=torch.int32, generator=gen)
        neg = torch.randint(1, V, (B, n), device=dev, dtype=torch.int32, generator=gen)
        batches.append([seq, pos, neg])
        real += int(lens.sum().item())

    def step(i):
        m(batches[i % NB])

    need = B * ((S + 1 + n) * 4 + (1 + n) * d * 4 + (1 + n) * 4) + (real / NB) * d * 4
    return {"step": step, "units": B, "work": need, "dtype": "f32", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "kernel": "sasrec_last_row_kernel (rec_sasrec_last_row_f32): the whole forward in one launch, exact last-row "
                      "form, item rows of real positions read once", "pmc_key": "sasrec_last_row_kernel",
            "workload": "SASRec seq 200 dim 64, 10M-item tables, 1 block, 100 negatives, global batch 8192 "
                        "(BASELINE configs[4])" + (", tables row-sharded over the ranks (RCCL all-to-all)"
                                                   if a.placement == "rowshard" else ""),
            "config": {"batch_per_gpu": B, "global_batch": B * world, "seq_len": S, "neg_len": n, "d": d,
                       "vocab_per_table": V, "mean_real_positions": round(real / NB / B, 2),
                       "bytes_note": "item rows of real positions + pos/neg rows + ids + logits"},
            "side_gather": None, "shape_cfg": {"workload": "sasrec", "batch": B, "vocab": V, "ids": a.ids},
            "sharded": getattr(m, "_sharded", None)}


SIDE_WORKLOADS = ("dlrm_fused", "gather", "sasrec")


def spawn_side_leg(a):
    """At N > 1 the OTHER table placement is measured in the same run and reported beside the headline — in a CHILD
    process per rank, started before this process touches the GPU and parked on its stdin until the headline is done.
    A collective that hangs or a communicator that aborts then costs the `rowshard` object, never the headline line.
    The children rendezvous among themselves on their own port."""
    import subprocess
    alt = "rowshard" if a.placement == "replicated" else "replicated"
    argv, skip = [], False
    for tok in sys.argv[1:]:
        if skip:
            skip = False
        elif tok == "--placement":
            skip = True
        elif not tok.startswith("--placement="):
            argv.append(tok)
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 137)
    for k in [k for k in env if k.startswith("TORCHELASTIC_") or k.startswith("TORCH_NCCL_ASYNC")]:
        env.pop(k)
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--placement", alt, "--side-leg", "--no-side", "--cpu-seconds", "0"]
    proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
    return alt, proc


def collect_side_leg(alt, proc, timeout):
    """Release the parked child, wait for its line (rank 0's child prints one), kill exactly that child on timeout."""
    import subprocess
    try:
        out, err = proc.communicate("go\n", timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"placement": alt, "error": f"timed out after {timeout:.0f} s"}
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        tail = (err or "").strip().splitlines()[-3:]
        return {"placement": alt, "error": f"side process rc={proc.returncode}: " + " | ".join(tail)[:400]}
    r = json.loads(lines[-1])
    return {"placement": alt, "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
            "launch_us": r["roofline"]["launch_us"], "exchange": r["config"].get("exchange")}


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: THIS process never touches a GPU (no torch import, no HIP call);
    it starts the N rank processes — the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, one GPU each — relays rank 0's JSON line and exits with the worst return code.  A rank that dies takes
    the others down (their exact PIDs); --launch-timeout bounds the whole run."""
    import socket
    import subprocess
    import threading
    n = a.gpus
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), REC_BENCH_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    deadline = time.time() + a.launch_timeout
    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = max(abs(c) for c in codes)
            if rc:
                why = "rank exit codes " + str(codes)
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.time() > deadline:
            why = f"rank {bad[0][0]} exited with {bad[0][1]}" if bad else f"no result after {a.launch_timeout:.0f} s"
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            rc = 1
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out = [ln for ln in lines if ln.startswith("{")]
    if out:
        print(out[-1].rstrip(), flush=True)
    if why:
        print(f"[bench] multi-GPU launch failed: {why}", file=sys.stderr, flush=True)
    return rc if (rc or out) else 1


def init_ranks(torch, dist, a, world, rank, local_rank):
    """device + process group of one rank.  RCCL ("nccl") is the backend; if its communicator cannot be created on
    this node the CONTROL PLANE (barrier, max-over-ranks of the times: the replicated headline has no data-path
    collective) falls back to gloo on a fresh port, and the line says so (config.control_backend)."""
    from datetime import timedelta
    # rehearsal knobs (one-GPU boxes only): REC_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # REC_BENCH_BACKEND=gloo replaces RCCL, to exercise the multi-rank control flow without N GPUs
    if os.environ.get("REC_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank}, this node exposes {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("REC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=180))
                dist.barrier()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: RCCL process group failed ({type(e).__name__}: {str(e)[:200]}); control plane "
                      "falls back to gloo", file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo (RCCL process group could not be created)"
                port = int(os.environ.get("MASTER_PORT", "29500")) + 61
                dist.init_process_group("gloo", init_method=f"tcp://{os.environ['MASTER_ADDR']}:{port}", rank=rank,
                                        world_size=world, timeout=timedelta(seconds=180))
        else:
            dist.init_process_group(backend, timeout=timedelta(seconds=180))
    return dev, backend


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.side_leg:
        sys.exit(launch_ranks(a))           # before any GPU call: the rank processes are fresh
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    side = None
    if a.side_leg:
        if not sys.stdin.readline().startswith("go"):     # parked until the parent has printed-ready its headline
            return
    elif world > 1 and not a.no_side and a.workload in SIDE_WORKLOADS:
        side = spawn_side_leg(a)
    import torch
    import torch.distributed as dist

    if world != a.gpus:
        a.gpus = world                      # under a launcher the environment is authoritative
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    dev, control_backend = init_ranks(torch, dist, a, world, rank, local_rank)
    if a.force:
        from recamd._lib import C
        for kv in a.force:
            k, _, v = kv.partition("=")
            C.debug_force(k, v)

    def barrier():
        if world > 1:
            dist.barrier()

    transport_note = {}

    def build(wl, placement):
        a.placement = placement
        if placement == "rowshard" and world > 1 and a.transport is None:
            # the library's own RCCL communicator first; if it cannot be created here, the same exchange over
            # torch.distributed's collectives (every rank takes the same branch: the failure modes are per-node) —
            # and the line SAYS so (config.exchange.transport / .fallback)
            try:
                return build_(wl)
            except Exception as e:  # noqa: BLE001
                transport_note["fallback"] = f"C-ABI RCCL transport failed ({type(e).__name__}: {str(e)[:160]})"
                print(f"[bench] rank {rank}: {transport_note['fallback']}; using torch.distributed", file=sys.stderr, flush=True)
                a.transport = "torch"
        return build_(wl)

    def build_(wl):
        if wl == "dlrm_fused":
            return wl_dlrm(torch, dev, a, rank, world, fused=True)
        if wl == "gather":
            return wl_dlrm(torch, dev, a, rank, world, fused=False)
        if wl == "autoint":
            return wl_autoint(torch, dev, a, rank, world)
        if wl == "din":
            return wl_din(torch, dev, a, rank, world)
        return wl_sasrec(torch, dev, a, rank, world)

    def spin(step, seconds):
        """untimed: run steps for `seconds` of wall time so clocks / TLBs are in their steady state"""
        t0, i = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(16):
                step(i)
                i += 1
            torch.cuda.synchronize()

    def timed(step, steps):
        """barrier + sync, `steps` launches bracketed by HIP events on the launch stream, sync + barrier;
        returns (wall seconds, mean device ms per step from the events)."""
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for i in range(steps):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        return t1 - t0, e0.elapsed_time(e1) / steps

    def per_launch(step, steps):
        """separate pass: every step between its own pair of events -> sorted per-step microseconds"""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(steps):
            step(i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps)]

    def stats(us):
        s = sorted(us)
        rest = us[1:] if len(us) > 1 else us
        return {"p10": round(s[len(s) // 10], 1), "p50": round(s[len(s) // 2], 1), "p90": round(s[(9 * len(s)) // 10], 1),
                "first": round(us[0], 1), "mean_without_first": round(sum(rest) / len(rest), 1), "n": len(us)}

    def measure(w, steps=None, warmup=None, spinup=None):
        steps = a.steps if steps is None else steps
        spin(w["step"], a.spinup if spinup is None else spinup)
        for i in range(a.warmup if warmup is None else warmup):
            w["step"](i)
        wall, dev_ms = timed(w["step"], steps)
        if world > 1:
            t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        launch = stats(per_launch(w["step"], min(steps, 200)))
        return wall, dev_ms, launch

    def roofline(w, dev_ms, launch):
        scale = 1e9 if w["unit"] == "GB/s" else 1e12
        ach = w["work"] / (dev_ms * 1e-3) / scale
        ach50 = w["work"] / (launch["p50"] * 1e-6) / scale
        traffic, src = (pmc_traffic(w["pmc_key"], w["shape_cfg"]) if w.get("pmc_key") and w.get("sharded") is None
                        else (None, None))
        key = "algorithmic_bytes_per_launch" if w["unit"] == "GB/s" else "flop_per_step"
        return {"kernel": w["kernel"], "bound": w["bound"], "achieved": round(ach, 1), "peak": w["peak"],
                "unit": w["unit"], "frac": round(ach / w["peak"], 4), "traffic": traffic, "traffic_source": src,
                key: int(w["work"]), "ms_per_launch": round(dev_ms, 4), "launch_us": launch,
                "frac_p50": round(ach50 / w["peak"], 4)}

    w = build(a.workload, a.placement)
    head = {k: w[k] for k in ("units", "dtype", "workload", "config")}
    wall, dev_ms, launch = measure(w)
    roof = roofline(w, dev_ms, launch)
    default_run = (world == 1 and not a.no_side and a.workload == "dlrm_fused" and w.get("sharded") is None
                   and a.ids == "uniform")

    # ---- side measurements in the same run -------------------------------------------------------------------------
    # the north-star gather kernel on the same tables
    gather_roof = None
    if not a.no_side and w.get("side_gather"):
        sg, sbytes = w["side_gather"]
        spin(sg, 0.1)
        _, g_ms = timed(sg, max(20, a.steps // 4))
        g_launch = stats(per_launch(sg, max(20, a.steps // 4)))
        ach = sbytes / (g_ms * 1e-3) / 1e9
        g_traffic, g_src = pmc_traffic("gather_uniform_kernel", w["shape_cfg"])
        gather_roof = {"kernel": "rec::gather_uniform_kernel<32, 0>", "bound": "hbm", "achieved": round(ach, 1),
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                       "frac_p50": round(sbytes / (g_launch["p50"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                       "traffic": g_traffic, "traffic_source": g_src, "algorithmic_bytes_per_launch": sbytes,
                       "ms_per_launch": round(g_ms, 4), "launch_us": g_launch,
                       "samples_per_s": round(w["units"] / (g_ms * 1e-3), 1)}

    def short(wx, steps, spinup=0.15):
        """one of the other measurements of the default line: same contract (spin-up, warm-up, `steps` timed launches
        between events, a per-launch pass), fewer steps"""
        wl_wall, wl_ms, wl_launch = measure(wx, steps=steps, warmup=5, spinup=spinup)
        r = roofline(wx, wl_ms, wl_launch)
        return {"workload": wx["workload"], "value": round(wx["units"] * steps / wl_wall, 1), "unit": "samples/s",
                "ms_per_step": round(wl_wall / steps * 1e3, 4), "steps": steps, "dtype": wx["dtype"],
                "launch_us": {k: wl_launch[k] for k in ("p10", "p50", "p90")},
                "roofline": {k: r[k] for k in r if k != "launch_us"}}

    # the headline step on Zipf(1.05) ids (both distributions are reported), same tables
    zipf = None
    if default_run:
        try:
            a.ids = "zipf"
            zids = make_ids(torch, dev, a, w["units"], a.fields, w["shape_cfg"]["vocab"], rank)
            a.ids = "uniform"
            from recamd import ops
            wz = dict(w, step=lambda i: ops.gather_pairwise_dot(w["group"], zids[i % NB], w["dense"], out=w["out_fused"]),
                      shape_cfg=dict(w["shape_cfg"], ids="zipf"),
                      workload=w["workload"] + ", Zipf(1.05) ids ((z - 1) mod V)")
            zipf = short(wz, max(20, a.steps // 2))
            shift = torch.arange(a.fields, device=dev, dtype=torch.int64)[None, :] * w["shape_cfg"]["vocab"]
            zipf["unique_rows_per_launch"] = int(torch.unique(zids[0].to(torch.int64) + shift).numel())
            del zids, wz
        except Exception as e:  # noqa: BLE001
            a.ids = "uniform"
            zipf = {"error": f"{type(e).__name__}: {e}"[:300]}

    # the measured table placement beside the plain allocation of the headline
    placed = None
    if default_run and a.placed_candidates > 1 and a.arena_candidates <= 1:
        try:
            placed = placed_side(torch, dev, a, w, short)
        except Exception as e:  # noqa: BLE001
            placed = {"error": f"{type(e).__name__}: {e}"[:300]}

    cpu_base = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0 and a.workload in ("dlrm_fused", "gather"):
        cpu_base = cpu_baseline(a, w)

    # PCIe-inclusive rate (never `value`): raw tokens + raw dense features start in pinned HOST memory; the
    # double-buffered feeder moves batch i+1 over PCIe and hashes / scales it on the device while batch i computes
    pcie = None
    if not a.no_side and world == 1 and a.workload == "dlrm_fused" and w.get("sharded") is None:
        try:
            pcie = pcie_inclusive(torch, dev, a, w, spin)
        except Exception as e:  # noqa: BLE001
            pcie = {"error": f"{type(e).__name__}: {e}"[:300]}

    exchange = None
    if w.get("sharded") is not None:
        exchange = w["sharded"].describe()
        exchange.update(transport_note)

    # the whole DLRM model around the headline step (same tables, same id batches): bottom MLP 13-512-256-128, the fused gather
    # + pairwise dot, top MLP 479-1024-1024-512-256-1 — what a served forward costs, next to the sparse stage it is built on
    model_forward = None
    if default_run and a.workload == "dlrm_fused" and a.placement == "replicated":
        try:
            w["headline_ms"] = wall / a.steps * 1e3
            model_forward = whole_model_forward(torch, dev, a, w)
        except Exception as e:  # noqa: BLE001
            model_forward = {"error": f"{type(e).__name__}: {e}"[:300]}

    # BASELINE configs[2..4] under the same contract, each with its own roofline (the DLRM tables are released first)
    configs = None
    if default_run:
        del w
        torch.cuda.empty_cache()
        configs = {}
        for name in ("autoint", "din", "sasrec"):
            try:
                a.batch, a.vocab = 0, 0
                wx = build_(name)
                configs[name] = short(wx, 100)
                configs[name]["config"] = wx["config"]
                del wx
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                configs[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        w = None

    res = None
    if rank == 0:
        res = {
            "metric": "forward samples/sec, Criteo-shape 65536x26 sparse x dim128",
            "value": round(world * head["units"] * a.steps / wall, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(wall / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": head["dtype"],
            "data": "synthetic",
            "config": dict({"workload": head["workload"], "step": a.workload, "placement": a.placement,
                            "parallelism": f"dp{world}", "spinup_s": a.spinup}, **head["config"]),
            "roofline": roof,
        }
        if world > 1:
            res["config"]["control_backend"] = control_backend
            res["config"]["launched_by"] = os.environ.get("REC_BENCH_LAUNCHED_BY", "external launcher (torch.distributed.run)")
        if exchange is not None:
            res["config"]["exchange"] = exchange
        for key, val in (("gather_roofline", gather_roof), ("zipf", zipf), ("placed", placed), ("model_forward", model_forward),
                         ("configs", configs),
                         ("cpu_baseline", cpu_base), ("pcie_inclusive", pcie)):
            if val is not None:
                res[key] = val

    # At N > 1: release the parked child processes (spawn_side_leg) once every rank has finished the headline and
    # returned its tables to the allocator; ranks other than 0 only wait for their child to end.
    if side is not None:
        w = None
        torch.cuda.empty_cache()
        barrier()
        other = collect_side_leg(side[0], side[1], a.side_timeout)
        if rank == 0:
            res[side[0]] = other

    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def placed_side(torch, dev, a, w, short):
    """`placed`: the headline step on the best of --placed-candidates arenas (recamd.ops.place_table_arena), the arena of
    the headline being candidate 0.  What a deployment that can afford the probe gains; never `value`."""
    from recamd import ops
    F, D, V = a.fields, a.dim, w["shape_cfg"]["vocab"]
    ids, dense, out = w["ids"], w["dense"], w["out_fused"]
    out_g = torch.empty((w["units"], F * D), dtype=torch.float32, device=dev)
    p_fused = lambda g, i: ops.gather_pairwise_dot(g, ids[i % NB], dense, out=out)  # noqa: E731
    p_gather = lambda g, i: ops.gather_concat(g, ids[i % NB], out=out_g)  # noqa: E731
    arena, info = ops.place_table_arena(F, V, D, dev, candidates=a.placed_candidates, probe=[p_fused, p_gather],
                                        probe_name="[rec_gather_pairwise_dot_f32, rec_gather_concat_f32] steps of this "
                                                   "workload, normalised times added", first=w["arena"])
    if info.get("chosen", 0) != 0:
        arena.copy_(w["arena"])                # same table contents as the headline
    group = ops.TableGroup([arena[f] for f in range(F)])
    wp = dict(w, step=lambda i: ops.gather_pairwise_dot(group, ids[i % NB], dense, out=out))
    r = short(wp, max(20, a.steps // 2))
    r.pop("workload")
    r["table_placement"] = info
    return r


def pcie_inclusive(torch, dev, a, w, spin):
    from recamd import ops
    from recamd.pipeline import BatchFeeder, MinMaxScaler
    B, F = w["ids"][0].shape
    nd, nb = 13, 48
    V = w["shape_cfg"]["vocab"]
    gen = torch.Generator().manual_seed(3)
    tok = torch.randint(-2 ** 31, 2 ** 31 - 1, (nb * B, F), dtype=torch.int32, generator=gen).pin_memory()
    raw = (torch.rand((nb * B, nd), generator=gen) * 1000).pin_memory()
    sc = MinMaxScaler().fit(raw[:B].to(dev))
    group = ops.TableGroup([w["arena"][f] for f in range(F)])
    dense = w["dense"]
    out = torch.empty((B, 480), dtype=torch.float32, device=dev)[:, :479]
    feeder = BatchFeeder(raw, tok, B, scaler=sc, hash_vocab=[V] * F, device=dev)
    for _d, ids in feeder:                      # untimed pass: pinned pages touched, clocks up
        ops.gather_pairwise_dot(group, ids, dense, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _d, ids in feeder:
        ops.gather_pairwise_dot(group, ids, dense, out=out)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    h2d = B * (F * 4 + nd * 4)
    return {"value": round(nb * B / el, 1), "unit": "samples/s", "ms_per_step": round(el / nb * 1e3, 4), "steps": nb,
            "h2d_bytes_per_step": h2d, "h2d_GBs": round(h2d * nb / el / 1e9, 2),
            "path": "pinned host tokens (B,26) uint32 + raw dense (B,13) fp32 -> H2D on a copy stream -> rec_hash_ids_u32 + "
                    "rec_minmax_scale_f32 on the device -> fused gather + pairwise dot; double-buffered (recamd.pipeline.BatchFeeder)"}


def whole_model_forward(torch, dev, a, w, steps=30):
    """DLRM (src/ctr/dlrm/model.py:42-54, the cited paper's dot interaction) end to end on the headline's tables: the mirror is
    built with one-row tables which are then replaced by views of the arena (no second 13 GB of tables)."""
    from ctr.dlrm.model import DLRM
    from recamd import ops
    arena, ids = w["arena"], w["ids"]
    F, V, D = arena.shape
    B = ids[0].shape[0]
    nd = 13
    sparse = [{'feat': f'C{i}', 'feat_num': 1, 'embed_dim': D} for i in range(F)]
    m = DLRM([[{'feat': f'I{i}'} for i in range(nd)], sparse], [512, 256, D], [1024, 1024, 512, 256], interaction='dot')
    for f in range(F):
        layer = m.embed_layers['embed_%d' % f]
        layer._w["embeddings"], layer.input_dim = arena[f], V
    m._group = ops.TableGroup([m.embed_layers['embed_%d' % f].table for f in range(F)])
    dense = torch.rand((B, nd), device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    for i in range(5):
        m([dense, ids[i % len(ids)]])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        m([dense, ids[i % len(ids)]])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    # the same forward with every Dense on the fp32-MFMA kernel (an fmaf chain; rec_debug_force is the A/B hook): how far the
    # 16-bit-matrix-core schemes are from plain fp32 arithmetic on this model's outputs (sigmoid probabilities)
    from recamd._lib import C
    sl = slice(0, min(B, 8192))
    got = m([dense[sl], ids[0][sl]]).reshape(-1)
    C.debug_force("dense", "f")
    try:
        ref32 = m([dense[sl], ids[0][sl]]).reshape(-1)
    finally:
        C.debug_force("dense", None)
    diff = float((got - ref32).abs().max().item())
    return {"max_abs_diff_vs_fp32_mfma_dense": diff, "compared_samples": int(sl.stop),
            "model": "DLRM dot: bottom MLP 13-512-256-128, gather + pairwise dot (26 x 1M x 128 tables), top MLP 479-1024-1024-512-256-1, "
                     "sigmoid; fp32 weights and activations, Dense layers on the f16x2 / bf16x3 kernels (fp32-accurate)",
            "batch": B, "steps": steps, "ms_per_forward": round(ms, 4), "value": round(B / ms * 1e3, 1), "unit": "samples/s",
            "sparse_stage_share": round(w.get("headline_ms", 0.0) / ms, 3) if w.get("headline_ms") else None}


def cpu_baseline(a, w):
    """The CPU restatement (oracle/oracle_c.c, OpenMP, kind "port") of the same workload on the box's host cores:
    the reference's own CPU path cannot run (TensorFlow is not installed, the reference never travels to the GPU
    box).  Bounded sample: slices of --cpu-samples samples, walked through ALL the id batches of the run (a fresh slice
    every pass: the looked-up rows come from host memory, not from a cache warmed by the previous pass), against the same
    13.3 GB tables copied to host memory, until the time budget is spent.  `torch_ops` beside it: the op sequence the
    reference executes — per-field index_select, cat, bmm, triangle gather, cat — as unfused torch CPU ops on
    all host threads, same slices, a quarter of the budget."""
    import numpy as np
    try:
        from oracle import c_oracle
        c_oracle.load()
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "samples/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    arena, dense = w["arena"], w["dense"]
    F, D = arena.shape[0], arena.shape[2]
    B = w["ids"][0].shape[0]
    Bc = min(a.cpu_samples, B)
    host = arena.cpu().numpy()  # (F, V, D) fp32
    tables = [host[f] for f in range(F)]
    # every slice of every id batch, in order: [batch][slice]
    slices = [(idb[lo:lo + Bc].cpu().numpy(), dense[lo:lo + Bc].cpu().numpy())
              for idb in w["ids"] for lo in range(0, B - Bc + 1, Bc)]
    n = F + 1
    scratch = np.empty((Bc, n * D), np.float32)
    out = np.empty((Bc, n * (n - 1) // 2 + D), np.float32)
    if a.workload == "dlrm_fused":
        fn = lambda s: c_oracle.dlrm_gather_dot(tables, s[0], s[1], scratch, out)  # noqa: E731
    else:
        fn = lambda s: c_oracle.gather_concat(tables, s[0])  # noqa: E731
    fn(slices[-1])  # warm (code and page tables; the timed passes start at slice 0)
    budget = a.cpu_seconds * 0.75
    t0 = time.perf_counter()
    reps = 0
    while True:
        fn(slices[reps % len(slices)])
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget or reps >= 1000:
            break
    res = {"value": round(Bc * reps / el, 1), "unit": "samples/s", "cores": c_oracle.num_threads(),
           "kind": "port",
           "sample": f"{reps} passes of {Bc} samples each ({el:.1f} s), walking the {len(slices)} distinct slices of the run's "
                     f"{len(w['ids'])} id batches in order ({Bc * F * D * 4 / 1e6:.0f} MB of looked-up rows per pass, a fresh "
                     f"slice every pass), same 26x1Mx128 tables in host memory, C/OpenMP restatement of "
                     f"gather+concat+pairwise-dot (TensorFlow unavailable)"}
    try:
        res["torch_ops"] = cpu_torch_ops(a, host, slices, a.cpu_seconds * 0.25)
    except Exception as e:  # noqa: BLE001
        res["torch_ops"] = {"value": None, "sample": f"unavailable: {e}"}
    return res


def cpu_torch_ops(a, host, slices, budget):
    """the same slices through the unfused op sequence of src/ctr/dlrm/model.py:45-50 (+ the cited paper's interaction) on
    torch CPU ops, all host threads"""
    import torch
    F, V, D = host.shape
    tabs = [torch.from_numpy(host[f]) for f in range(F)]
    n = F + 1
    il = torch.tril_indices(n, n, -1)        # the product's pair order: (i, j), i > j, rows = [fields ..., dense]

    def step(s):
        ids = torch.from_numpy(s[0]).long()
        emb = torch.cat([tabs[f].index_select(0, ids[:, f]) for f in range(F)], dim=-1)           # :45 gather + concat
        if a.workload != "dlrm_fused":
            return emb
        d = torch.from_numpy(s[1])
        T = torch.cat([emb, d], dim=-1).view(-1, n, D)
        Z = torch.bmm(T, T.transpose(1, 2))
        return torch.cat([Z[:, il[0], il[1]], d], dim=-1)
    step(slices[-1])
    t0 = time.perf_counter()
    reps = 0
    while True:
        step(slices[reps % len(slices)])
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget or reps >= 1000:
            break
    Bc = slices[0][0].shape[0]
    return {"value": round(Bc * reps / el, 1), "unit": "samples/s", "cores": torch.get_num_threads(),
            "sample": f"{reps} passes of {Bc} samples ({el:.1f} s): index_select per field, cat, bmm, triangle gather, cat as "
                      f"separate torch CPU ops (fp32), the same slices"}


if __name__ == "__main__":
    main()
