#!/usr/bin/env python3
"""bench.py -- headline benchmark: cosine top-100 queries/s over a 10M x 768 bf16 corpus resident in HBM.

A "step" is one 64-query batch searched against this rank's corpus shard (BASELINE.json configs[2]:
"1xMI355X: 10Mx768 bf16 corpus resident in HBM, batch-64 queries").  With N ranks every rank holds its
own 10M-row shard (weak scaling, configs[3]); a step additionally all-gathers the N local top-100
lists over RCCL and merges them on every rank.  `value` counts (query, 10M-row shard) scans per second
over all ranks, i.e. row.query pairs/s / 1e7.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus 8            # launches its own 8 ranks (torch.distributed.run) before touching the GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...    (the same thing, launched outside)

Prints ONE JSON line (rank 0).  Beside the headline it carries sub-records, each with its own roofline / parity block:
`filtered`, `wide` (one 512-query call), `config5` (scan + side-column gather + device hybrid re-rank, checked against the host
HybridRanker in-run), `f32_store` (ids and scores bit-exact vs the f32 oracle), `embed` (BASELINE configs[1], encoder only),
`c2` (configs[1] as one pipeline: 100k chunks -> packed encoder -> device-to-device index -> top-100 over the EMBEDDED vectors,
bit-exact vs the oracle on those vectors, and end-to-end recall against the fp32 pipeline on two weight statistics),
`embed_e2e` (texts -> provider -> python float lists), `c1` (BASELINE configs[0] shape: ~1k code chunks through the
reference-shaped surfaces, GPU vs the CPU oracle), `cpu_baseline`.  No reference code or CPU path is inside a timed GPU
region; the CPU legs (oracle = test infrastructure) run afterwards on rank 0 at N=1 only.

`--backend gloo` is the launcher rehearsal for machines without a GPU (tests/test_bench_launch.py): the same self-launch,
rendezvous, exchange-record all-gather, max-over-ranks timing and JSON assembly, with NO device work and `value` null.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_T0 = time.perf_counter()

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
ALL_LEGS = ("filtered", "scan_bf16", "wide", "config5", "clustered", "anisotropic", "f32_store", "embed", "c2", "embed_e2e", "index_e2e", "c1", "cpu")


def log(msg: str) -> None:
    """Progress on stderr (the JSON line on stdout stays alone); also keeps long runs visibly alive."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """Threads for the CPU baseline legs: the box's CPU share for one GPU is 16 cores even when more are visible."""
    return max(1, min(16, os.cpu_count() or 1))


def pct(a) -> dict:
    import numpy as np
    return {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90))}


# ------------------------------------------------------------------------------------------------ launcher
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no rendezvous in the environment: start N ranks (one per GPU) through
    torch.distributed.run as a CHILD process and hand its exit code back.  The parent never touches HIP (torch is not even
    loaded there), and nothing is exec'd from a process that has initialised the GPU."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    log(f"self-launch: {n} ranks via torch.distributed.run on 127.0.0.1:{port}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    return subprocess.run(cmd, env=env).returncode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per GPU")
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--seed-tiles", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--embed-chunks", type=int, default=100000, help="synthetic chunks for the encoder leg (BASELINE configs[1]: 100k synthetic code chunks)")
    ap.add_argument("--e2e-texts", type=int, default=20000, help="texts of the embed_e2e leg")
    ap.add_argument("--c2-parity-chunks", type=int, default=2000, help="chunks of the c2 leg's end-to-end recall subsample (per weight statistic)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --rows per GPU (BASELINE C4a); strong: --rows in total, split over the ranks (C4b)")
    ap.add_argument("--check-rows", type=int, default=1_000_000,
                    help="rows of the parity subsample checked vs the oracle (from 1M rows the sub-index is nominated from its int8 copy, like the timed corpus)")
    ap.add_argument("--legs", default=None, help="comma list of sub-records to run beside the headline "
                    f"({','.join(ALL_LEGS)}); default: all at N=1, config5+embed at N>1; 'none' = headline only")
    ap.add_argument("--sub-steps", type=int, default=20, help="timed steps of the filtered / f32_store / config5 legs")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="time box of each CPU baseline leg")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL (the measurement); gloo = launcher rehearsal without device work")
    ap.add_argument("--share-gpu", action="store_true",
                    help="with --backend gloo on a box with ONE GPU: every rank uses device 0 and the real N>1 path runs with "
                         "host-staged gloo collectives (rehearsal of everything but RCCL; not a measurement)")
    return ap.parse_args()


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    # stdout carries exactly ONE line, the JSON: libraries that print there (RCCL writes a version banner at communicator
    # creation) are sent to stderr for the whole run, and the line is written to the saved descriptor at the end
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.backend == "gloo" and not args.share_gpu:
        rehearse(args, json_fd)
    else:
        run(args, json_fd)


class HostStagedCollectives:
    """`torch.distributed` for device tensors over gloo (collectives staged through host memory): what `--backend gloo
    --share-gpu` uses to run the REAL N>1 path -- shard scans, merge kernel, side-column completion -- with several ranks on
    ONE GPU, where RCCL refuses to put two ranks on a device.  A rehearsal of everything but RCCL itself, not a measurement."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def all_gather_into_tensor(self, out, inp):
        o, i = out.cpu(), inp.cpu()
        self._d.all_gather_into_tensor(o, i)
        out.copy_(o)

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self._d.all_reduce(h, op=op or self._d.ReduceOp.SUM)
        t.copy_(h)

    def barrier(self):
        self._d.barrier()

    def get_world_size(self):
        return self._d.get_world_size()

    def destroy_process_group(self):
        self._d.destroy_process_group()


# ------------------------------------------------------------------------------------------------ the line on stdout
LINE_LIMIT = 4096      # bytes: the driver keeps a bounded tail of stdout + stderr; round 4's 22 KB line was cut and went unparsed
DETAIL_NAME = "bench_detail.json"


def _r(v, sig=6):
    """Floats at `sig` significant digits (the sidecar keeps full precision); everything else unchanged."""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{sig}g}")
    return v


def _all_true(d):
    """AND over every boolean found in a (nested) parity record; None when there is none."""
    seen = []

    def walk(o):
        if isinstance(o, bool):
            seen.append(o)
        elif isinstance(o, dict):
            for x in o.values():
                walk(x)
    walk(d)
    return all(seen) if seen else None


def _leg_short(name, rec):
    """One scalar + one fraction + one parity flag per sub-record."""
    if not isinstance(rec, dict):
        return None
    if "error" in rec:
        return {"error": str(rec["error"])[:80]}
    roof = rec.get("roofline") or {}
    unit = str(rec.get("unit") or "")
    s = {"value": _r(rec.get("value")), "unit": "chunks/s" if unit.startswith("chunks") else "texts/s" if unit.startswith("texts") else "q/s"}
    if rec.get("ms_per_step") is not None:
        s["ms"] = _r(rec["ms_per_step"], 5)
    if roof.get("frac") is not None:
        s["frac"] = _r(roof["frac"], 4)
        s["bound"] = roof.get("bound")
    if roof.get("whole_step_frac") is not None:
        s["step_frac"] = _r(roof["whole_step_frac"], 4)
    ok = _all_true({k: rec.get(k) for k in ("parity", "identical_to_headline")})
    if ok is not None:
        s["ok"] = ok
    if name == "c2":
        e2e = rec.get("end_to_end_vs_fp32_pipeline") or {}
        s["recall100_vs_fp32"] = {k: _r(v.get("recall_at_100"), 4) for k, v in e2e.items() if isinstance(v, dict)}
        s["recall100_residual_f32"] = {k: _r(((v.get("opt_in_levers") or {}).get("residual_f32") or {}).get("recall_at_100"), 4)
                                       for k, v in e2e.items() if isinstance(v, dict)}
        s["search_ms"] = _r((rec.get("search") or {}).get("ms_per_batch"), 4)
    if name == "c1":
        s = {"value": _r((rec.get("embed") or {}).get("value")), "unit": "chunks/s",
             "cpu": _r((rec.get("cpu_baseline") or {}).get("value"), 4),
             "min_cos": _r((rec.get("parity") or {}).get("min_cosine_gpu_vs_fp32_oracle"), 5)}
    if name == "config5":
        s["rerank_ms"] = _r(rec.get("rerank_ms_per_batch"), 4)
    if name == "index_e2e":
        s["reference_flow"] = _r((rec.get("sequential") or {}).get("value"), 5)
    return s


SHORT_LEGS = ("filtered", "scan_bf16", "clustered", "anisotropic", "f32_store", "wide", "config5", "embed", "c2", "embed_e2e", "index_e2e", "c1")


def contract_line(out: dict) -> dict:
    """What goes to stdout: the contract keys, the roofline block in both accountings, cpu_baseline, the headline's parity flags
    and one small record per leg.  Everything else is in the sidecar (`bench_detail.json`)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype")
    line = {k: _r(out.get(k)) for k in keep}
    line["data"] = "synthetic"
    cfg = out.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "rows_per_gpu", "dim", "batch", "k", "parallelism", "hbm_resident_bytes_per_row") if k in cfg}
    roof = out.get("roofline")
    if isinstance(roof, dict):
        r = {k: _r(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "launches",
                                         "algorithmic_bytes_per_launch", "step_bytes", "whole_step_frac")}
        b = roof.get("bf16_scan")
        if isinstance(b, dict):
            r["bf16_scan"] = {k: _r(b.get(k)) for k in ("kernel", "kernel_ms", "achieved", "frac", "algorithmic_bytes_per_launch", "ms_per_step",
                                                      "whole_step_frac", "identical_to_headline")}
        line["roofline"] = r
    else:
        line["roofline"] = None
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict) and "error" not in cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": str(cb.get("sample", ""))[:160]}
    else:
        line["cpu_baseline"] = cb if cb is None else {"error": str(cb.get("error"))[:80]}
    par = out.get("parity")
    if isinstance(par, dict):
        line["parity"] = {k: _r(par.get(k)) for k in ("rows", "ids_bit_exact", "scores_bit_exact", "nomination", "recall_at_k_vs_oracle_same_precision",
                                                       "merged_equals_sorted_concat_of_gathered_lists") if k in par}
    else:
        line["parity"] = None
    sd = out.get("step_ms_device")
    if isinstance(sd, dict):
        line["step_ms_device"] = {k: _r(v, 5) for k, v in sd.items()}
    st = out.get("search_stats") or {}
    line["fallback_used"] = st.get("fallback_used")
    for name in SHORT_LEGS:
        if name in out:
            line[name] = _leg_short(name, out[name])
    if out.get("n_gpus", 1) > 1 or out.get("backend"):
        for k in ("backend", "rccl_ranks", "collective_ranks", "per_rank_nomination", "per_rank_fallback_used", "exchange_layout_ok", "legs_rehearsed",
                  "rows_per_gpu", "rehearsal"):
            if k in out:
                line[k] = out[k]
        for k in ("per_rank_step_ms_device", "per_rank_step_ms", "per_rank_exchange_ms_device"):
            if out.get(k) is not None:
                line[k] = [_r(v, 5) for v in out[k]]
        dv = out.get("per_rank_device")
        if dv:
            line["per_rank_pci_bus_id"] = [(d or {}).get("pci_bus_id") for d in dv]
            line["per_rank_local_rank"] = [(d or {}).get("local_rank") for d in dv]
        ex = out.get("exchange_ms_device")
        if isinstance(ex, dict):
            line["exchange_ms_device"] = {k: _r(v, 5) for k, v in ex.items() if k != "what"}
    if out.get("errors"):
        line["errors"] = [str(e)[:120] for e in out["errors"]][:6]
    line["detail"] = DETAIL_NAME
    return line


def write_all(fd: int, data: bytes) -> None:
    """os.write may write less than it was given (a pipe): loop until every byte is out."""
    view = memoryview(data)
    while view:
        n = os.write(fd, view)
        view = view[n:]


def emit(out: dict, json_fd: int) -> None:
    """Sidecar first (full record, next to bench.py and -- when that scratch directory exists -- under gpurun_out/), then the ONE
    short strict-JSON line on stdout.  A line over LINE_LIMIT sheds its per-leg records rather than going out unparseable."""
    try:
        full = json.dumps(out, allow_nan=False)
    except ValueError:            # a NaN / inf somewhere in a sub-record: the sidecar keeps it as null
        full = json.dumps(json.loads(json.dumps(out), parse_constant=lambda c: None), allow_nan=False)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_NAME), "w") as f:
                    f.write(full + "\n")
            except OSError as e:
                log(f"sidecar {d}/{DETAIL_NAME} not written: {e!r}")
    line = contract_line(out)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for name in reversed(SHORT_LEGS):
        if len(text) + 1 <= LINE_LIMIT:
            break
        if name in line:
            line[name] = {"value": (line[name] or {}).get("value")} if isinstance(line[name], dict) else None
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) + 1 > LINE_LIMIT:
        raise SystemExit(f"bench line is {len(text) + 1} bytes, over the {LINE_LIMIT}-byte contract")
    write_all(json_fd, (text + "\n").encode())
    log(f"line {len(text) + 1} B on stdout; full record ({len(full)} B) in {DETAIL_NAME}")


# ------------------------------------------------------------------------------------------------ rehearsal (no GPU)
def rehearse(args, json_fd) -> None:
    """The N>1 control flow on CPU: rendezvous, ONE all-gather of the [scores | rows] records per step (the buffers and views
    ffi.topk_exchange_buffers hands the real path), barrier + max-over-ranks timing, per-rank gathers, rank 0 prints the
    line.  The local top-k lists are synthetic and nothing is searched or merged: this is NOT a measurement."""
    import torch
    import torch.distributed as dist
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    import datetime
    # (a rank that dies mid-run must take the others down in bounded time, not after gloo's default 30 minutes)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("CODERAG_BENCH_COLLECTIVE_TIMEOUT_S", "120"))))
    B, K = args.queries, args.k
    dev = torch.device("cpu")
    local, loc_s, loc_r, gathered, all_s, all_r = ffi.topk_exchange_buffers(torch, world, B, K, dev)
    n_local = args.rows // world if args.scaling == "strong" else args.rows
    layout_ok = True

    def step(i):
        loc_s.copy_(torch.arange(B * K, dtype=torch.float32).view(B, K) * -1.0 - rank - i)
        loc_r.copy_(torch.arange(B * K, dtype=torch.int64).view(B, K) + rank * n_local)
        dist.all_gather_into_tensor(gathered.view(-1), local)

    for i in range(args.warmup):
        step(i)
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    dist.barrier()
    dt = time.perf_counter() - t0
    for r in range(world):        # rank r's record landed in slot r, both halves
        layout_ok = layout_ok and float(all_s[r, 0, 0]) == float(-r - (args.steps - 1)) and int(all_r[r, 0, 1]) == 1 + r * n_local
    # the collectives of the legs an N > 1 run carries beside the headline (config5: the all-gather above + ONE all-reduce that
    # completes the packed side columns -- every candidate is owned by exactly one rank, the others contribute zeros; embed: no
    # data-path collective, barriers + max-over-ranks timing), with their layouts checked
    legs = set(("config5", "embed", "c2") if args.legs is None else (x for x in args.legs.split(",") if x and x != "none"))
    legs_ok = {}
    if "config5" in legs:
        n = B * K
        words = n * (6 + ffi.RR_NAME_BYTES // 4)                       # ranking.device.PackedColumns: six int32 columns + 64 name bytes
        cand = torch.arange(n, dtype=torch.int64)                       # candidate c is owned by rank c % world
        packed = torch.zeros((words,), dtype=torch.int32)
        mine = (cand % world) == rank
        for c in range(6):
            packed[c * n:(c + 1) * n][mine] = (cand[mine] * 7 + c).to(torch.int32)
        packed[6 * n:].view(n, ffi.RR_NAME_BYTES // 4)[mine] = (cand[mine, None] + torch.arange(ffi.RR_NAME_BYTES // 4)[None, :]).to(torch.int32)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        ok = all(bool(torch.equal(packed[c * n:(c + 1) * n], (cand * 7 + c).to(torch.int32))) for c in range(6))
        ok = ok and bool(torch.equal(packed[6 * n:].view(n, -1), (cand[:, None] + torch.arange(ffi.RR_NAME_BYTES // 4)[None, :]).to(torch.int32)))
        legs_ok["config5"] = bool(ok)
    if "embed" in legs:
        dist.barrier()
        te = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        legs_ok["embed"] = bool(abs(float(te.item()) - 0.001 * world) < 1e-12)
    if "c2" in legs:      # per-rank embed -> local shard (no collective) -> local top-k -> ONE all-gather of the records -> merge; parity flags meet in one all-reduce(MIN)
        step(0)
        flag = torch.tensor([1], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        legs_ok["c2"] = bool(int(flag.item()) == 1 and all(int(all_r[r, 0, 1]) == 1 + r * n_local for r in range(world)))
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    devices = [None] * world                              # (the real run fills these from crh_device_info: a mis-pinned rank shows)
    dist.all_gather_object(devices, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_name": None, "hbm_bytes": None})
    ranks = torch.zeros((world,), dtype=torch.int64)
    dist.all_gather_into_tensor(ranks, torch.tensor([rank], dtype=torch.int64))
    per_rank = torch.zeros((world,), dtype=torch.float64)
    dist.all_gather_into_tensor(per_rank, torch.tensor([dt * 1e3 / max(1, args.steps)], dtype=torch.float64))
    # how every rank's batches were nominated and what fell back (the real run: crh_index_get_nomination / crh_search_stats of
    # each rank's shard): [mode, fallback_used] per rank in one all-gather -- here a recognisable stand-in, -1 - rank / rank
    nom = torch.zeros((world, 2), dtype=torch.int64)
    dist.all_gather_into_tensor(nom.view(-1), torch.tensor([-1 - rank, rank], dtype=torch.int64))
    errs = first_contact_errors(world, int(len(set(ranks.tolist()))), devices, None, None)   # (devices here: distinct local ranks)
    if rank == 0:
        out = {"metric": "launcher rehearsal (gloo, CPU): no device work, not a measurement", "value": None, "unit": None,
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) * 1e3 / max(1, args.steps),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": None, "data": "synthetic records",
               "config": {"workload": f"rehearsal: {world} gloo ranks, {B}x{K} exchange records, {n_local} rows per rank"},
               "backend": "gloo", "collective_ranks": int(len(set(ranks.tolist()))), "rccl_ranks": None,
               "exchange_layout_ok": bool(layout_ok), "per_rank_step_ms": [float(v) for v in per_rank.tolist()],
               "rows_per_gpu": n_local, "legs_rehearsed": legs_ok, "per_rank_device": devices,
               "per_rank_nomination": [int(v) for v in nom[:, 0].tolist()], "per_rank_fallback_used": [int(v) for v in nom[:, 1].tolist()]}
        if errs:
            out["errors"] = errs
        emit(out, json_fd)
    dist.destroy_process_group()
    if not layout_ok or not all(legs_ok.values()) or errs:
        raise SystemExit(f"rehearsal: exchange layout ok = {layout_ok}, legs = {legs_ok}, first-contact checks = {errs}")


def first_contact_errors(world, collective_ranks, per_rank_device, per_rank_nom, merge_ok, shared_gpu=False) -> list:
    """What an N > 1 record must show before its number means anything: N distinct ranks took part in the collectives, no two of
    them sat on the same GPU, every rank nominated its batches the way rank 0 did (a rank that fell back to the bf16 tiles
    halves the job's rate without failing anything), and the merged list equals the sort of the gathered ones."""
    errs = []
    if collective_ranks != world:
        errs.append(f"{collective_ranks} distinct ranks took part in the all-gather, --gpus {world}")
    if per_rank_device and not shared_gpu:
        seen = {}
        for d in per_rank_device:
            bus = (d or {}).get("pci_bus_id")
            key = bus if bus is not None else ("local_rank", (d or {}).get("local_rank"))
            if key in seen:
                errs.append(f"ranks {seen[key]} and {(d or {}).get('rank')} report the same device ({key})")
            seen[key] = (d or {}).get("rank")
    if per_rank_nom:
        for r, (mode, fb) in enumerate(per_rank_nom):
            if mode != per_rank_nom[0][0]:
                errs.append(f"rank {r} nominated its batches in mode {mode}, rank 0 in mode {per_rank_nom[0][0]}")
            if fb & 6:
                errs.append(f"rank {r} fell back during the timed steps (fallback bits {fb})")
    if merge_ok is False:
        errs.append("the merged top-k differs from the sort of the gathered lists")
    return errs


# ------------------------------------------------------------------------------------------------ the measurement
CORPUS_KINDS = ("gaussian", "clustered", "anisotropic")


def corpus_generator(torch, dev, kind, seed, D=768):
    """Row blocks of one of the corpus kinds SURVEY 8(d) / the round-3 review name, generated on the device:
    gaussian     rows ~ N(0, I) (the headline; scores ~ N(0, 1/D));
    clustered    1000 unit centres + 0.7 * N(0, I)/sqrt(D) (SURVEY 8(d)'s secondary corpus); queries: a centre + the same noise;
    anisotropic  encoder-like: x = c + a * P z / sqrt(r) + b * g / sqrt(D) with ONE shared unit direction c, a 64-dimensional
                 noise subspace P (orthonormal) and a little isotropic noise, a^2 + b^2 = 0.352 -- the mean unit vector then
                 has norm 1/sqrt(1.352) = 0.86 and a query's top-1 and top-100 of 10M rows lie ~0.03 apart, the statistics
                 `c2.embedding_geometry` reports for encoder output; queries: fresh draws of the same model.
    Returns (block(m) -> [m, D] f32 tensor, queries(n, seed) -> [n, D] f32 ndarray)."""
    import numpy as np
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    if kind == "gaussian":
        return (lambda m: torch.randn((m, D), generator=gen, device=dev, dtype=torch.float32),
                lambda n, qseed: np.random.default_rng(qseed).standard_normal((n, D)).astype(np.float32))
    if kind == "clustered":
        centres = torch.randn((1000, D), generator=gen, device=dev)
        centres /= centres.norm(dim=1, keepdim=True)

        def block(m):
            pick = torch.randint(0, 1000, (m,), generator=gen, device=dev)
            return centres[pick] + 0.7 * torch.randn((m, D), generator=gen, device=dev) / (D ** 0.5)

        def queries(n, qseed):
            r = np.random.default_rng(qseed)
            return (centres.cpu().numpy()[r.integers(0, 1000, n)] + 0.7 * r.standard_normal((n, D)) / np.sqrt(D)).astype(np.float32)
        return block, queries
    if kind == "anisotropic":
        r0 = np.random.default_rng(seed)
        rank_, a2, b2 = 64, 0.302, 0.05
        c = r0.standard_normal(D)
        c /= np.linalg.norm(c)
        P = np.linalg.qr(r0.standard_normal((D, rank_)))[0].astype(np.float32)            # [D, 64], orthonormal columns
        c_d, P_d = torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(P).to(dev)

        def block(m):
            z = torch.randn((m, rank_), generator=gen, device=dev)
            g = torch.randn((m, D), generator=gen, device=dev)
            return c_d[None, :] + (a2 / rank_) ** 0.5 * (z @ P_d.T) + (b2 / D) ** 0.5 * g

        def queries(n, qseed):
            r = np.random.default_rng(qseed)
            return (c[None, :] + (a2 / rank_) ** 0.5 * (r.standard_normal((n, rank_)) @ P.T)
                    + (b2 / D) ** 0.5 * r.standard_normal((n, D))).astype(np.float32)
        return block, queries
    raise ValueError(kind)


def build_corpus(torch, ffi, dev, rows, dtype, seed, check_rows, code_cols=1, stream=0, seed_tiles=0, kind="gaussian"):
    """Rows of `kind` (corpus_generator) generated on the device in blocks (never staged through host lists), normalised on
    insert; one dictionary-coded payload column (`language`: 3 uniform codes) beside them, as every collection of the
    reference has keyword payload indexes (embeddings/client.py:77-89).  Returns (index, head rows, head codes) -- the head
    (the first check_rows rows) is the parity subsample."""
    D = 768
    idx = ffi.Index(D, dtype, capacity_rows=rows, n_code_cols=code_cols, device=dev.index)
    if seed_tiles:
        idx.set_tuning(seed_tiles=seed_tiles)
    block_of, _ = corpus_generator(torch, dev, kind, seed, D)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 1)
    heads, head_codes = [], []
    kept = 0
    block = 500_000
    for r0 in range(0, rows, block):
        m = min(block, rows - r0)
        xb = block_of(m)
        cb = torch.randint(0, 3, (m, code_cols), generator=gen, device=dev, dtype=torch.int32) if code_cols else None
        idx.append(xb, codes=cb, stream=stream)
        if kept < check_rows:
            t = min(check_rows - kept, m)
            heads.append(xb[:t].cpu().numpy())
            if cb is not None:
                head_codes.append(cb[:t].cpu().numpy())
            kept += t
        torch.cuda.synchronize()
        del xb, cb
    import numpy as np
    head = np.concatenate(heads) if heads else None
    return idx, head, (np.concatenate(head_codes) if head_codes else None)


NSLOTS = 4  # rotating output buffers


def run_steps(n, launch, complete):
    """n steps: every step completes on the launch stream before the next one starts."""
    for i in range(n):
        launch(i)
        complete(i)


def timed_search(torch, idx, qd, K, filters, steps, warmup, stream):
    """warmup + `steps` timed batches of one configuration: wall around a full synchronize, per-step device stamps on the
    launch stream (the completion of step i, joined to it), HIP-event time of the scan kernel from the library's own profiling."""
    import numpy as np
    nq, dev = qd.shape[0], qd.device
    out_s = [torch.empty((nq, K), dtype=torch.float32, device=dev) for _ in range(NSLOTS)]
    out_r = [torch.empty((nq, K), dtype=torch.int64, device=dev) for _ in range(NSLOTS)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]

    def launch(i):
        idx.search(qd, K, filters=filters, out_scores=out_s[i % NSLOTS], out_rows=out_r[i % NSLOTS], stream=stream)

    # sub-records only (the headline's W warm-up steps are counted by its own loop): at least `warmup` batches AND >= 60 ms of
    # device work -- a leg that follows seconds of CPU-side checking (the `filtered` leg comes right after the parity subsample)
    # otherwise starts on an idle chip's clocks: its 20 steps read 1.36 / 1.45 / 1.66 ms (p10 / median / p90) against 1.35 flat
    run_steps(warmup, launch, lambda i: None)
    idx.search_finish(stream)
    torch.cuda.synchronize()
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.06:
        run_steps(4, launch, lambda i: None)
        idx.search_finish(stream)
        torch.cuda.synchronize()
    idx.set_profiling(True)
    t0 = time.perf_counter()
    ev[0].record()
    run_steps(steps, launch, lambda i: ev[i + 1].record())
    idx.search_finish(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    scan_ms, launches = idx.profile()
    idx.set_profiling(False)
    per = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
    last = (steps - 1) % NSLOTS
    return {"ms_per_step": wall * 1e3 / steps, "step_ms_device": pct(per), "scan_ms": scan_ms / max(1, launches),
            "scan_launches": launches, "stats": idx.stats(), "scores": out_s[last], "rows": out_r[last]}


def kernel_source_sha16() -> str:
    """tools/summarize_prof.py stamps the counter pass with this fingerprint of the scan kernels' sources."""
    import hashlib
    h = hashlib.sha256()
    for f in ("crh_i8.hpp", "crh_kernels.hpp"):
        h.update(open(os.path.join(ROOT, "code-rag_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def scan_kernel_name(D, mode):
    """The dominant kernel of a <= 64-query batch, by the nomination mode the index reports (crh_index_get_nomination):
    2 = the pass over the int8 copy (the third of that scan's three launches; the sample tiles and the thresholds are the
    `<.., 1>` and `<.., 2>` launches before it), 1 = the one-launch scan over the bf16 tiles, 0 = the three-launch bf16 form."""
    qb = 1 if D == 1536 else 2
    if mode == 2:
        return f"k_scan_i8<{D // 32},8,8,{qb},3>"
    if mode == 1:
        return f"k_scan_fused<{D // 16},16,8,{qb}>"
    return f"k_scan<{D // 16},1,16,8,{qb}>"


def i8_sample_record() -> bool:
    """Whether the library's int8 scan records the sample tiles' upper ends (its default; CODERAG_HIP_I8_SAMPLE_RECORD=0 turns it off)."""
    return os.environ.get("CODERAG_HIP_I8_SAMPLE_RECORD", "1")[:1] != "0"


def scan_roofline(rows, D, scan_ms, launches, mode, stats=None):
    """Algorithmic bytes of one pass: the int8 copy is 1 byte per element + one f32 scale per row, the bf16 tiles 2 bytes per
    element; every mode reads its copy of the corpus once (the three-launch form's seed scan is outside the timed kernel).
    Since round 5 the int8 PASS does not read the sample tiles again (the sample launch before it recorded their rows' upper ends,
    32 bytes per lane and query block): its bytes are the other tiles + that record -- `algorithmic_bytes_per_launch` is what the
    timed kernel reads by design, `step_bytes` is every row once (the whole step: sample launch + pass)."""
    step_bytes = float(rows) * (D + 4) if mode == 2 else float(rows) * D * 2
    alg = step_bytes
    sample_tiles = None
    if mode == 2 and i8_sample_record() and stats and stats.get("batches"):
        sample_tiles = stats["seed_tiles"] / stats["batches"]
        alg = step_bytes - sample_tiles * 32 * (D + 4) + sample_tiles * 64 * (32 if D == 1536 else 64)
    ach = alg / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
           "kernel": scan_kernel_name(D, mode), "kernel_ms": scan_ms, "launches": launches, "algorithmic_bytes_per_launch": alg,
           "step_bytes": step_bytes, "sample_tiles_not_read_again": sample_tiles,
           "bytes_per_row": (D + 4) if mode == 2 else D * 2,
           "nomination": ("int8 copy of the rows (i8 MFMA, exact integer dot, per-row error intervals)" if mode == 2 else "bf16 tiles (bf16 MFMA)")
                         + "; every returned id and score: canonical f32 arithmetic on the stored rows"}
    if mode == 2:
        # the scan is three launches since late round 4 (no grid-wide wait); the timed kernel is the third.  The first two --
        # 8192 sample tiles (~200 MB read again by the pass) and the thresholds -- are outside `kernel_ms` and inside
        # `ms_per_step` / `whole_step_frac`, like the query preparation and the selection (profiles/: kernel trace of this command)
        out["kernel_is"] = ("the pass over the copy: the third of the scan's three launches (k_scan_i8<..,1> sample tiles and <..,2> thresholds run before it, not in "
                            "kernel_ms); it takes the sample tiles' candidates from the record the first launch left and reads the other tiles")
    return out


def subsample_parity(np, ffi, orc, head, head_codes, qs, K, dtype, device, filters=None, truth=True):
    """HIP index over the first rows of the corpus vs the oracle on the same rows: ids and f32 score bits.  The record says
    which scan nominated the rows of THIS check (`nomination` / `kernel`: from 1M rows up the library's default is the int8 copy,
    the path of the 10M-row line) and what that search did (`search_stats`)."""
    bf16 = dtype == ffi.DTYPE_BF16
    ncols = 0 if head_codes is None else head_codes.shape[1]
    sub = ffi.Index(head.shape[1], dtype, capacity_rows=head.shape[0], n_code_cols=ncols, device=device)
    sub.append(head, codes=head_codes)
    gs, gr = sub.search(qs, K, filters=filters)
    mode, st = sub.nomination(), sub.stats()
    sub.close()
    kw = {"codes": head_codes, "filters": list(filters)} if filters else {}
    es, er = orc.cosine_search(head, qs, K, bf16=bf16, **kw)
    ts, tr = orc.cosine_search(head, qs, K, bf16=False, **kw) if truth else (None, None)

    def recall(a, b):
        return float(np.mean([len(set(x[x >= 0]) & set(y[y >= 0])) / max(1, int((y >= 0).sum())) for x, y in zip(a, b)]))
    return {"rows": int(head.shape[0]), "queries": int(qs.shape[0]), "ids_bit_exact": bool(np.array_equal(gr, er)),
            "scores_bit_exact": bool(np.array_equal(gs.view(np.uint32), es.view(np.uint32))),
            "nomination": {2: "int8 copy", 1: "bf16 tiles, one launch", 0: "bf16 tiles, three launches"}.get(mode, str(mode)),
            "kernel": scan_kernel_name(head.shape[1], mode),
            "search_stats": {k: st[k] for k in ("candidates", "max_query_cands", "fallback_used") if k in st},
            "recall_at_k_vs_oracle_same_precision": recall(gr, er), "recall_at_k_vs_f32_truth": recall(gr, tr) if truth else None,
            "max_abs_score_error_vs_f32_truth": float(np.max(np.abs(np.sort(gs, axis=1) - np.sort(ts, axis=1)))) if truth else None}


def run(args, json_fd) -> None:
    import numpy as np
    import torch
    import coderag_amd  # noqa: F401
    from coderag_amd import ffi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # CODERAG_BENCH_FORCE_DIST=1 runs the N>1 code path (RCCL init, all-gather, merge, max-over-ranks) with one rank -- the
    # rehearsal available on a one-GPU box
    force_dist = os.environ.get("CODERAG_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("CODERAG_BENCH_COLLECTIVE_TIMEOUT_S", "600")))
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
            dist = HostStagedCollectives(dist)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
    legs = set(ALL_LEGS if world == 1 else ("config5", "embed", "c2")) if args.legs is None else \
        set(x for x in args.legs.split(",") if x and x != "none")
    unknown = legs - set(ALL_LEGS)
    if unknown:
        raise SystemExit(f"unknown legs: {sorted(unknown)}")
    if args.no_cpu_baseline or world > 1:
        legs -= {"cpu"}
    if world > 1:
        legs -= {"c1", "embed_e2e", "index_e2e", "f32_store", "filtered", "scan_bf16", "wide", "clustered", "anisotropic"}      # one-GPU verification legs

    D, N, B, K = 768, args.rows, args.queries, args.k
    if args.scaling == "strong":
        N = (args.rows + world - 1) // world
    dtype = ffi.DTYPE_BF16 if args.dtype == "bf16" else ffi.DTYPE_F32
    stream = torch.cuda.current_stream().cuda_stream

    idx, head, head_codes = build_corpus(torch, ffi, dev, N, dtype, 20251226 + rank, args.check_rows, 1, stream, args.seed_tiles)
    log(f"corpus resident: {N} rows x {D} ({args.dtype})")
    qs = np.random.default_rng(7).standard_normal((B, D)).astype(np.float32)
    qd = torch.from_numpy(qs).to(dev)
    row_base = rank * N

    # per-rank result records [scores | rows] (ffi.topk_exchange_buffers): the exchange is ONE all-gather per step
    nslots = NSLOTS
    slots = [ffi.topk_exchange_buffers(torch, world, B, K, dev) for _ in range(nslots)]
    out_s, out_r = [sl[1] for sl in slots], [sl[2] for sl in slots]
    mer_s = mer_r = None
    if dist is not None:
        mer_s = torch.empty((B, K), dtype=torch.float32, device=dev)
        mer_r = torch.empty((B, K), dtype=torch.int64, device=dev)

    # per-step device time stamps on the launch stream (torch's current stream): step i spans ev[i] .. ev[i+1]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ev_x = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)] if dist is not None else None

    timing = [False]

    def launch(i: int) -> None:
        idx.search(qd, K, row_base=row_base, out_scores=out_s[i % nslots], out_rows=out_r[i % nslots], stream=stream)

    def complete(i: int) -> None:
        if dist is not None:
            if timing[0]:
                ev_x[i].record()
            local, _, _, gathered, gat_s, gat_r = slots[i % nslots]
            dist.all_gather_into_tensor(gathered.view(-1), local)
            ffi.merge_topk(gat_s, gat_r, mer_s, mer_r, stream)
        if timing[0]:
            ev[i + 1].record()

    def fence() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup, launch, complete)
    idx.search_finish(stream)
    fence()
    idx.set_profiling(True)
    timing[0] = True
    t0 = time.perf_counter()
    ev[0].record()
    run_steps(args.steps, launch, complete)
    idx.search_finish(stream)  # also verifies no candidate buffer overflowed in any timed step
    fence()
    dt = time.perf_counter() - t0
    timing[0] = False
    log(f"search timed: {args.steps} steps in {dt:.3f} s")
    per_step = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)])
    exchange = np.array([ev_x[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]) if dist is not None else None
    scan_ms_total, scan_launches = idx.profile()
    idx.set_profiling(False)
    stats = idx.stats()
    last = (args.steps - 1) % nslots
    headline_rows = out_r[last].clone()          # this rank's local top-k of the last timed step (global row ids)
    headline_scores = out_s[last].clone()
    nom_mode = idx.nomination()                  # how the timed batches were nominated (2: int8 copy, 1 / 0: bf16 tiles)
    # outside the timed region: the same batches once more with the library's two events around ALL three launches of the int8
    # scan (sample tiles, thresholds, pass) -- the span that was one launch, and `kernel_ms`, until late in round 4
    scan3_ms = None
    if nom_mode == 2:
        idx.set_profiling(2)
        run_steps(min(20, args.steps), launch, lambda i: None)   # (rank-local: no collective -- the ranks need not agree on the mode)
        idx.search_finish(stream)
        torch.cuda.synchronize()
        t3, n3 = idx.profile()
        idx.set_profiling(False)
        scan3_ms = t3 / max(1, n3)

    # ---- what actually took part in the exchange (N>1): distinct ranks seen through a real all-gather, per-rank step times,
    # and the merged list against a sort of the gathered lists (score descending, lower global row first)
    rccl_ranks, per_rank_ms, merge_ok, per_rank_device, per_rank_nom, per_rank_exchange_ms = None, None, None, None, None, None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ids = torch.zeros((dist.get_world_size(),), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
        rccl_ranks = int(len(set(ids.tolist())))
        pr = torch.zeros((dist.get_world_size(),), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(pr, torch.tensor([float(np.median(per_step))], dtype=torch.float64, device=dev))
        per_rank_ms = [float(v) for v in pr.tolist()]
        px = torch.zeros((dist.get_world_size(),), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(px, torch.tensor([float(np.median(exchange))], dtype=torch.float64, device=dev))
        per_rank_exchange_ms = [float(v) for v in px.tolist()]
        nomt = torch.zeros((dist.get_world_size(), 2), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(nomt.view(-1), torch.tensor([int(nom_mode), int(stats["fallback_used"])], dtype=torch.int64, device=dev))
        per_rank_nom = [[int(a), int(b)] for a, b in nomt.tolist()]
        info = ffi.device_info(local_rank)
        per_rank_device = [None] * dist.get_world_size()
        me = {"rank": rank, "local_rank": local_rank, "device_name": info["name"], "arch": info["arch"], "hbm_bytes": info["hbm_bytes"],
              "cu_count": info["cu_count"], "pci_bus_id": torch.cuda.get_device_properties(local_rank).pci_bus_id
              if hasattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id") else None}
        (dist._d if isinstance(dist, HostStagedCollectives) else dist).all_gather_object(per_rank_device, me)
        _, _, _, _, gat_s, gat_r = slots[last]
        cat_s = gat_s.permute(1, 0, 2).reshape(B, -1)
        cat_r = gat_r.permute(1, 0, 2).reshape(B, -1)
        o1 = torch.argsort(cat_r, dim=1, stable=True)                                   # by row, then (stable) by score
        o2 = torch.argsort(torch.gather(cat_s, 1, o1), dim=1, descending=True, stable=True)
        order = torch.gather(o1, 1, o2)[:, :K]
        merge_ok = bool(torch.equal(torch.gather(cat_r, 1, order), mer_r) and torch.equal(torch.gather(cat_s, 1, order), mer_s))

    # ---- parity on a subsample (outside the timed region): first check_rows rows of rank 0 vs the oracle
    parity = None
    orc = None
    if rank == 0 and head is not None and args.check_rows > 0:
        from oracle import search as orc
        parity = subsample_parity(np, ffi, orc, head, head_codes, qs, K, dtype, local_rank)
        if merge_ok is not None:
            parity["merged_equals_sorted_concat_of_gathered_lists"] = merge_ok

    # HBM traffic of the scan kernel comes from PMC counters, which need their own rocprofv3 passes (tools/gpu_prof.sh);
    # the corrected per-launch figure of the committed pass is reported when it was taken at this corpus size
    traffic, traffic_note = None, "no counter pass on record (profiles/pmc_scan.json)"
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_scan.json")))
        if int(pmc.get("rows", -1)) != N or pmc.get("kernel") != scan_kernel_name(D, nom_mode):
            traffic_note = f"the counter pass on record is of {pmc.get('kernel')} at {pmc.get('rows')} rows, not of this run's kernel / size"
        elif pmc.get("kernel_source_sha16") != kernel_source_sha16():
            traffic_note = "the kernel sources have changed since the counter pass on record was taken (tools/gpu_prof.sh takes a new one)"
        else:
            traffic = float(pmc["hbm_read_bytes_corrected"]) + float(pmc["hbm_write_bytes"])
    except (OSError, ValueError, KeyError):
        pass
    ms_per_step = dt * 1e3 / args.steps
    value = world * B * args.steps / dt * (N / 1e7)
    scan_ms = scan_ms_total / max(1, scan_launches)
    roof = scan_roofline(N, D, scan_ms, scan_launches, nom_mode, stats)
    roof["traffic"] = traffic
    roof["whole_step_frac"] = roof["step_bytes"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS   # the whole step: every row's bytes once
    if scan3_ms:
        roof["scan_three_launches_ms"] = scan3_ms
        roof["scan_three_launches_frac"] = roof["step_bytes"] / (scan3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["traffic_source"] = ("profiles/pmc_scan.json (separate rocprofv3 --pmc passes of this kernel's sources; FETCH_SIZE x2 per the gfx950 guide)"
                              if traffic else traffic_note)
    out = {
        "metric": "top-k queries/s over 10Mx768 (cosine top-100, batch 64)",
        "value": value,
        "unit": "queries/s per 10M-row shard (row.query pairs/s / 1e7)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "bf16" if args.dtype == "bf16" else "f32",
        "data": "synthetic: N(0,I) rows generated on device (torch seed 20251226+rank), normalised on insert; queries numpy default_rng(7)",
        "config": {"workload": f"{world}x MI355X: {N}x{D} {args.dtype} corpus per GPU resident in HBM, batch-{B} queries, exact top-{K}"
                               + (", all-gather + merge of per-GPU top-k over RCCL" if world > 1 else "")
                               + (f" ({args.rows} rows in total, strong scaling)" if args.scaling == "strong" else ""),
                   "rows_per_gpu": N, "dim": D, "batch": B, "k": K, "parallelism": f"row-shard x{world}",
                   "precision": ("bf16 corpus; rows nominated from " + ("an int8 copy (i8 MFMA, exact integer dot, rigorous per-row error intervals)"
                                                                              if nom_mode == 2 else "the bf16 tiles (bf16 MFMA, f32 accumulate)")
                                 + ", canonical f32 re-score of the survivors on the bf16 rows decides every id and score" if args.dtype == "bf16"
                                 else "f32 store: the scan nominates, f32 canonical re-score decides"),
                   "hbm_resident_bytes_per_row": D * (2 if args.dtype == "bf16" else 6) + (D + 4 if nom_mode == 2 else 0)},
        "roofline": roof,
        "step_ms_device": pct(per_step),
        "per_rank_step_ms_device": per_rank_ms,
        "per_rank_exchange_ms_device": per_rank_exchange_ms,      # median of [all-gather + merge] per step, every rank's own clock
        "per_rank_device": per_rank_device,
        # (2 = every batch of that rank was nominated from its int8 copy; fallback bits: 1 buffers regrown, 2 a grid-wide wait timed out, 4 a batch went to the bf16 scan)
        "per_rank_nomination": [a for a, _ in per_rank_nom] if per_rank_nom else None,
        "per_rank_fallback_used": [b for _, b in per_rank_nom] if per_rank_nom else None,
        "rccl_ranks": rccl_ranks if args.backend == "nccl" else None,
        "collective_ranks": rccl_ranks, "backend": args.backend if dist is not None else None,
        "exchange_ms_device": (dict(pct(exchange), what="1 RCCL all-gather of the [scores | rows] records + k_merge_topk, rank 0")
                               if exchange is not None else None),
        "search_stats": stats,
        "parity": parity,
    }
    if args.backend == "gloo":
        out["rehearsal"] = "ranks share one GPU and the collectives are host-staged gloo: everything but RCCL itself; NOT a measurement"
    # first contact with a real N-GPU node: what must hold for the line to mean "N GPUs", checked by every rank on the gathered
    # records (so that all ranks leave the same way); the line still goes out, then the run exits non-zero
    first_contact = first_contact_errors(world, rccl_ranks, per_rank_device, per_rank_nom, merge_ok, shared_gpu=args.share_gpu) if dist is not None else []
    if first_contact:
        out["errors"] = first_contact
    log("parity subsample checked" if parity else "parity subsample skipped")

    def leg(name, fn, *a, key=None):
        """A failing sub-record must not cost the headline: it is reported as {"error": ...} and the run goes on."""
        if name not in legs:
            return
        name = key or name
        try:
            res = fn(*a)
            if rank == 0 and res is not None:
                out[name] = res
        except Exception as e:  # noqa: BLE001
            log(f"leg {name} FAILED: {e!r}")
            if rank == 0:
                out[name] = {"error": repr(e)}

    def filtered_leg():
        r = timed_search(torch, idx, qd, K, [(0, 1)], args.sub_steps, 3, stream)
        res = {"workload": f"{N}x{D} {args.dtype}, batch-{B} top-{K}, filter language == code 1 of 3 (uniform)",
               "value": B / (r["ms_per_step"] * 1e-3) * (N / 1e7), "unit": out["unit"], "ms_per_step": r["ms_per_step"],
               "steps": args.sub_steps, "step_ms_device": r["step_ms_device"],
               "roofline": scan_roofline(N, D, r["scan_ms"], r["scan_launches"], idx.nomination(), r["stats"]), "search_stats": r["stats"]}
        if orc is not None:
            m = min(200_000, head.shape[0])
            res["parity"] = subsample_parity(np, ffi, orc, head[:m], head_codes[:m], qs, K, dtype, local_rank, filters=[(0, 1)])
        log(f"filtered: {r['ms_per_step']:.3f} ms/step")
        return res

    def scan_bf16_leg():
        """The headline batch with the int8 copy switched off: the one-launch scan over the bf16 tiles (rounds 1-2's path, and
        what an index falls back to).  Same rows, same score bits."""
        if nom_mode != ffi.NOMINATE_INT8:
            return None
        idx.set_nomination(ffi.NOMINATE_BF16)
        try:
            r = timed_search(torch, idx, qd, K, None, args.sub_steps, 3, stream)
            mode = idx.nomination()
        finally:
            idx.set_nomination(ffi.NOMINATE_INT8)
        same = bool(torch.equal(r["rows"], headline_rows - row_base) and torch.equal(r["scores"].view(torch.int32), headline_scores.view(torch.int32)))
        log(f"scan_bf16: {r['ms_per_step']:.3f} ms/step")
        roof16 = scan_roofline(N, D, r["scan_ms"], r["scan_launches"], mode)
        # SURVEY 8(d)'s own accounting (N x 768 x 2 bytes per launch, the bf16 brute-force pass north_star's ">= 70 % of the HBM
        # roofline" speaks of) sits in the HEADLINE's roofline block beside the int8 pass's figure on the bytes IT reads
        out["roofline"]["bf16_scan"] = {"kernel": roof16["kernel"], "kernel_ms": roof16["kernel_ms"], "achieved": roof16["achieved"], "frac": roof16["frac"],
                                        "algorithmic_bytes_per_launch": roof16["algorithmic_bytes_per_launch"], "ms_per_step": r["ms_per_step"],
                                        "whole_step_frac": roof16["step_bytes"] / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "identical_to_headline": same,
                                        "what": "the same batch on the same index with the int8 copy switched off (crh_index_set_nomination): "
                                                "the one-launch scan over the bf16 tiles, 2 bytes per element"}
        return {"workload": f"{N}x{D} {args.dtype}, batch-{B} top-{K}, nominated from the bf16 tiles", "value": B / (r["ms_per_step"] * 1e-3) * (N / 1e7),
                "unit": out["unit"], "ms_per_step": r["ms_per_step"], "steps": args.sub_steps, "step_ms_device": r["step_ms_device"],
                "roofline": roof16, "search_stats": r["stats"],
                "identical_to_headline": same}

    def corpus_kind_leg(kind):
        """The headline batch on a corpus that is NOT N(0, I): `clustered` (SURVEY 8(d)'s secondary corpus) and `anisotropic`
        (encoder-like: one shared direction, low-rank noise -- corpus_generator).  What the int8 intervals nominate depends on the
        data: each record carries `search_stats` (candidates per query, fallbacks), the roofline on the bytes the pass read, the
        geometry of the corpus, and parity of a 1M-row sub-index (int8 copy live there too) against the oracle."""
        def run_leg():
            kidx, khead, kcodes = build_corpus(torch, ffi, dev, N, dtype, 777 + rank, args.check_rows, 1, stream, args.seed_tiles, kind=kind)
            try:
                _, queries_of = corpus_generator(torch, dev, kind, 777 + rank, D)
                kq = queries_of(B, 7)
                kqd = torch.from_numpy(kq).to(dev)
                r = timed_search(torch, kidx, kqd, K, None, args.sub_steps, 3, stream)
                mode = kidx.nomination()
                roof_k = scan_roofline(N, D, r["scan_ms"], r["scan_launches"], mode, r["stats"])
                roof_k["whole_step_frac"] = roof_k["step_bytes"] / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                st = r["stats"]
                gs = r["scores"].cpu().numpy()
                res = {"workload": f"{N}x{D} {args.dtype} {kind} corpus, batch-{B} top-{K}", "value": B / (r["ms_per_step"] * 1e-3) * (N / 1e7),
                       "unit": out["unit"], "ms_per_step": r["ms_per_step"], "steps": args.sub_steps, "step_ms_device": r["step_ms_device"],
                       "roofline": roof_k, "search_stats": st,
                       "candidates_per_query_and_batch": st["candidates"] / max(1, st["batches"]) / B if "batches" in st else None,
                       "fallback_used": st.get("fallback_used"),
                       "geometry": {"top1_score_median": float(np.median(gs[:, 0])), "topk_score_median": float(np.median(gs[:, K - 1])),
                                    "top1_minus_topk_median": float(np.median(gs[:, 0] - gs[:, K - 1]))}}
                if khead is not None:
                    hn = khead[:100_000] / np.linalg.norm(khead[:100_000], axis=1, keepdims=True)
                    res["geometry"]["norm_of_mean_unit_vector"] = float(np.linalg.norm(hn.mean(0)))
                if orc is not None and khead is not None:
                    res["parity"] = subsample_parity(np, ffi, orc, khead, kcodes, kq, K, dtype, local_rank, truth=False)
                log(f"{kind}: {r['ms_per_step']:.3f} ms/step, kernel {r['scan_ms']:.3f} ms, candidates/query {res['candidates_per_query_and_batch']}, "
                    f"fallback {st.get('fallback_used')}")
                return res
            finally:
                kidx.close()
                torch.cuda.empty_cache()
        return run_leg

    def wide_leg():
        """One search call with 512 queries: two passes of k_scan_wide (256 queries share a corpus pass, query fragments in
        registers, corpus tiles through an LDS-DMA ring) instead of eight 64-query passes.  MFMA-bound, not HBM-bound."""
        nq = 512
        q512 = np.random.default_rng(11).standard_normal((nq, D)).astype(np.float32)
        q512[:B] = qs
        qd512 = torch.from_numpy(q512).to(dev)
        r = timed_search(torch, idx, qd512, K, None, max(5, args.sub_steps // 2), 2, stream)
        flops = 2.0 * 256 * N * D
        ach = flops / (r["scan_ms"] * 1e-3) / 1e12 if r["scan_ms"] > 0 else 0.0
        same = bool(torch.equal(r["rows"][:B], headline_rows - row_base))
        # one 64-query slice of the batch searched on its own: the same rows, the same score bits
        s64 = torch.empty((64, K), dtype=torch.float32, device=dev)
        r64 = torch.empty((64, K), dtype=torch.int64, device=dev)
        idx.search(qd512[320:384], K, out_scores=s64, out_rows=r64, stream=stream)
        idx.search_finish(stream)
        torch.cuda.synchronize()
        same = same and bool(torch.equal(r64, r["rows"][320:384]) and torch.equal(s64.view(torch.int32), r["scores"][320:384].view(torch.int32)))
        res = {"workload": f"{N}x{D} {args.dtype}, ONE search call with {nq} queries, exact top-{K}", "value": nq / (r["ms_per_step"] * 1e-3) * (N / 1e7),
               "unit": out["unit"], "ms_per_step": r["ms_per_step"], "steps": max(5, args.sub_steps // 2), "step_ms_device": r["step_ms_device"],
               "passes_per_call": r["scan_launches"] / max(5, args.sub_steps // 2),
               "roofline": {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                            "kernel": "k_scan_wide<48,1>", "kernel_ms": r["scan_ms"], "launches": r["scan_launches"], "algorithmic_flops_per_launch": flops,
                            "hbm_GBps": float(N) * D * 2 / (r["scan_ms"] * 1e-3) / 1e9, "traffic": None},
               "search_stats": r["stats"],
               "parity": {"identical_to_64_query_passes": same, "what": "rows and f32 score bits of queries 0..63 (the headline batch) and 320..383"}}
        if orc is not None:
            m = min(50_000, head.shape[0])
            res["parity"].update(subsample_parity(np, ffi, orc, head[:m], head_codes[:m] if head_codes is not None else None, q512, K, dtype, local_rank))
        log(f"wide: {r['ms_per_step']:.3f} ms per 512-query call")
        return res

    leg("filtered", filtered_leg)
    leg("scan_bf16", scan_bf16_leg)
    leg("wide", wide_leg)
    leg("config5", config5_leg, np, torch, ffi, dist, idx, qd, N, B, K, rank, world, row_base, args.sub_steps, stream)
    idx.close()
    del idx
    torch.cuda.empty_cache()

    def f32_leg():
        f32, h32, hc32 = build_corpus(torch, ffi, dev, N, ffi.DTYPE_F32, 20251226 + rank, min(args.check_rows, 200_000), 1, stream, args.seed_tiles)
        try:
            r = timed_search(torch, f32, qd, K, None, args.sub_steps, 3, stream)
            a_, b_ = r["rows"].cpu().numpy(), (headline_rows - row_base).cpu().numpy()
            full = float(np.mean([len(np.intersect1d(x, y)) / K for x, y in zip(a_, b_)])) if args.dtype == "bf16" else None
            res = {"workload": f"{N}x{D} f32 store (bf16 MFMA scan nominates, f32 master re-scores), batch-{B} top-{K}",
                   "value": B / (r["ms_per_step"] * 1e-3) * (N / 1e7), "unit": out["unit"], "ms_per_step": r["ms_per_step"],
                   "steps": args.sub_steps, "step_ms_device": r["step_ms_device"],
                   "roofline": scan_roofline(N, D, r["scan_ms"], r["scan_launches"], f32.nomination(), r["stats"]), "search_stats": r["stats"],
                   "hbm_resident_bytes": float(N) * D * (7 if f32.nomination() == 2 else 6),
                   "recall_at_k_of_the_bf16_store_vs_this_store_full_corpus": full}
            if orc is not None:
                res["parity"] = subsample_parity(np, ffi, orc, h32, hc32, qs, K, ffi.DTYPE_F32, local_rank)
            log(f"f32 store: {r['ms_per_step']:.3f} ms/step")
            return res
        finally:
            f32.close()
            torch.cuda.empty_cache()

    leg("clustered", corpus_kind_leg("clustered"))
    leg("anisotropic", corpus_kind_leg("anisotropic"))
    leg("f32_store", f32_leg)
    if args.embed_chunks <= 0:
        legs.discard("embed")
    leg("embed", embed_leg, np, torch, local_rank, args.embed_chunks, rank, world, dist,
        ("cpu" in legs), args.cpu_seconds)
    if args.embed_chunks <= 0:
        legs.discard("c2")
    leg("c2", c2_leg, np, torch, ffi, local_rank, args.embed_chunks, rank, B, K, args.c2_parity_chunks if world == 1 else 0, dist, world)
    ckpt = {}
    leg("embed_e2e", embed_e2e_leg, np, torch, local_rank, args.e2e_texts, ckpt)
    leg("index_e2e", index_e2e_leg, np, torch, local_rank, ckpt)
    leg("c1", c1_leg, np, torch, local_rank, ckpt, args.cpu_seconds * 4 if "cpu" in legs else 0.0)
    if "cpu" in legs and rank == 0:
        leg("cpu", cpu_baseline, np, B, K, D, args.cpu_seconds, key="cpu_baseline")
        out["cpu_baseline_c1"] = (out.get("c1") or {}).get("cpu_baseline")
    if rank == 0:
        emit(out, json_fd)
    if dist is not None:
        dist.destroy_process_group()
    if first_contact:
        for e in first_contact:
            log(f"N>1 CHECK FAILED: {e}")
        raise SystemExit(3)


# ------------------------------------------------------------------------------------------------ config 5
def config5_leg(np, torch, ffi, dist, idx, qd, N, B, K, rank, world, row_base, steps, stream):
    """BASELINE config 5: vector top-k (all-gathered + merged over the ranks when N>1) -> side columns of the candidates
    gathered on the device (one all-reduce completes them across shards) -> crh_rerank_vector.  The output is compared
    IN-RUN with the host HybridRanker (itself pinned bit-for-bit to the reference's ranker by tests/golden/ranking_*.json)
    on hit dictionaries rebuilt from the same gathered columns.  Side arrays per SURVEY 8(d): content_len lognormal
    (seed 99), degree Zipf (seed 98), one plan per query cycling the QueryIntent weight rows."""
    from types import SimpleNamespace as NS
    from coderag_amd.engine_helpers import centrality_candidates
    from coderag_amd.query_types import GraphContext, QueryIntent
    from coderag_amd.ranking import HybridRanker
    from coderag_amd.ranking.device import SIGNALS, DeviceReranker, SideColumns
    dev = qd.device
    vocab = [f"fn_{i}" for i in range(2000)] + ["UserRepository", "verify_password", "parse_file", ""]
    # degree is a property of the graph NODE (= the centrality key): one Zipf draw per name, shared by all ranks
    r99, r98 = np.random.default_rng(99 + 1000 * rank), np.random.default_rng(98)
    name_id = r99.integers(0, len(vocab), N)
    names = np.zeros((len(vocab), 64), np.uint8)
    nlen = np.zeros(len(vocab), np.int32)
    for i, v in enumerate(vocab):
        b = v.lower().encode()
        names[i, :len(b)] = np.frombuffer(b, np.uint8)
        nlen[i] = len(b)
    grow = row_base + np.arange(N, dtype=np.int64)                                # global row ids of this shard
    side = SideColumns.from_arrays(dev.index, content_len=np.clip(np.round(np.exp(r99.normal(np.log(400.0), 1.0, N))), 0, 20000),
                                   degree=(np.minimum(r98.zipf(1.6, len(vocab)), 500) - 1)[name_id], file_code=(grow // 7 + 1).astype(np.int32),
                                   key_code=(grow + 1).astype(np.int32), node_code=name_id + 1, name_len=nlen[name_id], name=names[name_id])
    del grow
    intents = [i.value for i in QueryIntent]
    plans = [NS(primary_intent=intents[q % len(intents)], entities=[NS(name=vocab[(37 * q) % 2000]), NS(name="Repository")]) for q in range(B)]
    local, loc_s, loc_r, gathered, gat_s, gat_r = ffi.topk_exchange_buffers(torch, world, B, K, dev)
    s = torch.empty((B, K), dtype=torch.float32, device=dev)
    r = torch.empty((B, K), dtype=torch.int64, device=dev)
    rr = DeviceReranker(device=dev.index)
    multi = dist is not None and world > 1
    cols_box = {}

    def once():
        if multi:
            idx.search(qd, K, row_base=row_base, out_scores=loc_s, out_rows=loc_r, stream=stream)
            dist.all_gather_into_tensor(gathered.view(-1), local)
            ffi.merge_topk(gat_s, gat_r, s, r, stream)
        else:
            idx.search(qd, K, row_base=row_base, out_scores=s, out_rows=r, stream=stream)
        # (stream order carries the results on; crh_search_finish after the loop checks that no candidate buffer overflowed)
        cols = side.gather(r, row_base=row_base, stream=stream)
        if multi:
            dist.all_reduce(cols.packed, op=dist.ReduceOp.SUM)
        cols_box["cols"] = cols
        return rr.rank(s, r, cols, plans, stream=stream)     # ends with the D2H copy of the survivors: the step is complete

    for _ in range(3):
        out = once()
    idx.search_finish(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    # (a) one batch at a time: the next search is enqueued only after the previous batch's survivors have reached the host
    t0 = time.perf_counter()
    for _ in range(steps):
        out = once()
    idx.search_finish(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall_one = (time.perf_counter() - t0) / steps
    if idx.stats()["fallback_used"]:
        raise RuntimeError("a candidate buffer overflowed inside the timed config-5 steps")

    # (b) the serving loop: one stream, searches enqueued ahead of the re-rank they feed (three sets of buffers), and the NEXT search
    # enqueued between a re-rank's launches and the wait for its survivors -- so that the device never waits for the host's share
    # of a step (packing the plans, the launches, the copy back: ~0.25 ms).  Every step still ends with its batch's survivors on
    # the host.  The two searches in flight when the clock starts are complete by then (no head start); the two enqueued last are
    # waited for before it stops: `steps` searches and `steps` re-ranks.  (The re-rank on a stream of its own, waiting only for
    # its own search, was measured too: its eight small launches then each wait for a gap between the scans' chip-wide kernels,
    # 1.75 ms per step against 1.61 one batch at a time.)
    sets = [(s, r, (local, loc_s, loc_r, gathered, gat_s, gat_r))]
    for _ in range(2):
        sets.append((torch.empty_like(s), torch.empty_like(r), ffi.topk_exchange_buffers(torch, world, B, K, dev)))

    def enqueue_search(b):
        bs, br, (blocal, bloc_s, bloc_r, bgathered, bgat_s, bgat_r) = b
        if multi:
            idx.search(qd, K, row_base=row_base, out_scores=bloc_s, out_rows=bloc_r, stream=stream)
            dist.all_gather_into_tensor(bgathered.view(-1), blocal)
            ffi.merge_topk(bgat_s, bgat_r, bs, br, stream)
        else:
            idx.search(qd, K, row_base=row_base, out_scores=bs, out_rows=br, stream=stream)

    def enqueue_rerank(b):
        bs, br, _ = b
        cols = side.gather(br, row_base=row_base, stream=stream)
        if multi:
            dist.all_reduce(cols.packed, op=dist.ReduceOp.SUM)
        cols_box["cols"] = cols
        return rr.rank_async(bs, br, cols, plans, stream=stream)

    enqueue_search(sets[0])
    enqueue_search(sets[1])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    idx.set_profiling(True)
    t0 = time.perf_counter()
    for i in range(steps):
        collect = enqueue_rerank(sets[i % 3])            # batch i: behind search i+1, already enqueued
        enqueue_search(sets[(i + 2) % 3])                # search i+2: queued before the host starts to wait
        out = collect()                                  # batch i's survivors on the host: the step is complete
    idx.search_finish(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = (time.perf_counter() - t0) / steps
    if idx.stats()["fallback_used"]:
        raise RuntimeError("a candidate buffer overflowed inside the timed config-5 steps")
    scan_ms, launches = idx.profile()
    idx.set_profiling(False)
    s, r = sets[(steps - 1) % 3][0], sets[(steps - 1) % 3][1]      # the batch `out` belongs to (the parity check below rebuilds its hits)
    t0 = time.perf_counter()
    for _ in range(steps):
        rr.rank(s, r, cols_box["cols"], plans, stream=stream)
    rerank = (time.perf_counter() - t0) / steps
    if dist is not None:
        t = torch.tensor([wall, wall_one], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, wall_one = float(t[0].item()), float(t[1].item())
    sharded = store_sharded_check(np, torch, dist, dev, rank, world)
    if rank != 0:
        return None
    # parity + host cost: HybridRanker over hit dicts rebuilt from the gathered (and, for N>1, all-reduced) columns
    cols = {k: v.cpu().numpy() for k, v in cols_box["cols"].items()}
    rows_h, scores_h = r.cpu().numpy(), s.cpu().numpy()
    host = HybridRanker()
    ok, t_host, survivors = True, 0.0, 0
    for q in range(B):
        base = q * K
        hits, degs = [], []
        for j, (ro, sc) in enumerate(zip(rows_h[q], scores_h[q])):
            if ro < 0:
                continue
            c = base + j
            cl = int(cols["content_len"][c])
            hits.append({"score": float(sc), "file_path": f"f{int(cols['file_code'][c])}", "entity_name": vocab[int(cols["node_code"][c]) - 1],
                         "entity_type": "function", "graph_node_id": None, "content": "x" * cl if cl else None,
                         "start_line": int(ro), "end_line": 0})
            degs.append(int(cols["degree"][c]))
        cand = centrality_candidates(GraphContext(), hits)
        deg_of = {h["entity_name"]: d for h, d in zip(hits[:5], degs[:5])}
        table = {n: {"total_degree": deg_of[n]} for n in cand if deg_of.get(n, -1) >= 0}
        t1 = time.perf_counter()
        want = host.rank_results(plans[q], GraphContext(), hits, table)
        t_host += time.perf_counter() - t1
        got = DeviceReranker.materialise(out, q, hits)
        survivors += len(want)
        ok = ok and len(got) == len(want) and all(
            g.final_score == w.final_score and g.entity_name == w.entity_name and g.start_line == w.start_line and g.source == w.source
            and [g.signal_scores[n] for n in SIGNALS] == [w.signal_scores[n] for n in SIGNALS] for g, w in zip(got, want))
    scan = scan_ms / max(1, launches)
    log(f"config5: scan+exchange+rerank {wall * 1e3:.3f} ms/batch, rerank part {rerank * 1e3:.3f} ms, host ranker {t_host * 1e3:.1f} ms, parity {ok}")
    return {"workload": f"{world}x {N}x768 bf16, {B} queries, top-{K}" + (" all-gathered + merged" if multi else "")
                        + " -> side-column gather" + (" + all-reduce" if multi else "") + " -> device hybrid re-rank -> <= 50 per query"
                        + "; searches enqueued two batches ahead of the re-rank they feed, every step ends with its survivors on the host",
            "value": world * B / wall * (N / 1e7), "unit": "queries/s per 10M-row shard (row.query pairs/s / 1e7), re-ranked",
            "ms_per_step": wall * 1e3, "steps": steps, "ms_per_step_one_batch_at_a_time": wall_one * 1e3, "rerank_ms_per_batch": rerank * 1e3,
            "host_hybrid_ranker_ms_per_batch": t_host * 1e3, "survivors": survivors,
            "roofline": scan_roofline(N, 768, scan, launches, idx.nomination(), idx.stats()),
            "store_sharded": sharded,
            "parity": {"identical_to_host_hybrid_ranker": bool(ok), "queries": B,
                       "what": "survivors, order, f64 final scores, the four signals and the source label of every query"}}


def store_sharded_check(np, torch, dist, dev, rank, world, n=20000, nq=16):
    """The row-sharded index THROUGH the reference's store surface: HipVectorStore(shards=...) -- one process per GPU over the
    run's process group at N > 1, three in-process shards on the one GPU at N = 1 -- upsert of payload dictionaries, graph
    degrees, search_rerank_batch (per-shard scans -> [all-gather ->] crh_merge_topk_strided -> side columns summed over the
    shards -> crh_rerank_vector), compared with the same calls on an unsharded store and with the host HybridRanker."""
    import asyncio
    from coderag_amd.engine_helpers import search_and_rank_batch_device
    from coderag_amd.query_types import ExtractedEntity, QueryIntent, QueryPlan
    from coderag_amd.ranking import HybridRanker
    from coderag_amd.ranking.device import DeviceReranker
    from coderag_amd.store import HipVectorStore
    if dist is not None and isinstance(dist, HostStagedCollectives):
        return {"skipped": "gloo rehearsal on a shared GPU: the store's dist backend issues device collectives"}
    rng = np.random.default_rng(2025)                      # the same stream on every rank: replicated calls
    vecs = rng.standard_normal((n, 768)).astype(np.float32)
    pay = [{"file_path": f"/p/f{i % 500}.py", "entity_type": "function", "entity_name": f"fn_{i % 900}", "language": "python", "start_line": i % 37,
            "end_line": i % 37 + 5, "content": "x" * int(c), "graph_node_id": None if i % 3 else f"m.fn_{i % 900}", "content_hash": "h",
            "project_name": "p"} for i, c in enumerate(rng.integers(0, 3000, n))]
    qs = rng.standard_normal((nq, 768)).astype(np.float32)
    intents = list(QueryIntent)
    plans = [QueryPlan(f"q{i}", intents[i % len(intents)], entities=[ExtractedEntity(f"fn_{37 * i % 900}"), ExtractedEntity("fn")]) for i in range(nq)]
    multi = dist is not None and world > 1
    ns, backend = (world, "dist") if multi else (3, "local")

    async def go(shards, backend):
        async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=4096, device=dev.index, shards=shards, shard_backend=backend) as st:
            await st.create_collections()
            for a in range(0, n, 5000):
                await st.upsert("code_chunks", [f"id{i}" for i in range(a, a + 5000)], vecs[a:a + 5000], pay[a:a + 5000])
            await st.set_graph_degrees("code_chunks", {f"m.fn_{j}": j % 60 for j in range(0, 900, 2)})
            ranked = await search_and_rank_batch_device(st, DeviceReranker(device=dev.index), HybridRanker(), qs, plans, limit=20)
            info = await st.get_collection_info("code_chunks")
            return [[(r.file_path, r.entity_name, r.start_line, r.final_score, r.source, tuple(sorted(r.signal_scores.items()))) for r in per] for per in ranked], info
    t0 = time.perf_counter()
    got, info = asyncio.run(go(ns, backend))
    ref, _ = asyncio.run(go(1, "local"))
    return {"shards": ns, "backend": backend, "rows": n, "shard_rows": info.config["shard_rows"], "queries": nq,
            "identical_to_unsharded_store": bool(got == ref and all(len(p) > 0 for p in got)), "seconds": time.perf_counter() - t0,
            "what": "HipVectorStore(shards=N).search_rerank_batch vs the same calls on shards=1: survivors, order, f64 scores, signals, source"}


# ------------------------------------------------------------------------------------------------ encoder legs
def embed_leg(np, torch, local_rank, n_chunks, rank, world, dist, cpu, cpu_seconds):
    """Second half of BASELINE.json's metric: chunks embedded/s by the HIP UniXcoder encoder (configs[1] shape:
    synthetic chunks, lengths ~ clip(round(exp(N(ln 160, 0.8^2))), 8, 512), seeded random RoBERTa-base weights in bf16).
    Each rank embeds its own n_chunks (weak scaling, no collective).  FLOPs are counted on TRUE lengths."""
    from coderag_amd import encoder as drv
    cfg = drv.EncoderConfig()
    model = drv.HipUniXcoder(drv.synthetic_weights(cfg, 23), cfg, drv.HashTokenizer(cfg.vocab_size), local_rank)
    rng = np.random.default_rng(1234 + rank)
    lengths = np.clip(np.round(np.exp(rng.normal(np.log(160.0), 0.8, n_chunks))), 8, 512).astype(np.int64)
    dev = torch.device("cuda", local_rank)
    # packed batches (no padding: rows back to back on one token axis, attention / gather / pool take the row offsets):
    # what HipUniXcoder.embed_ids / embed_bodies submit; the ids are staged on the device before the timed region
    batches = []
    id_rows = synth_chunk_ids(np, cfg, lengths, rng)
    for rows, _ in model.plan_batches(lengths, max_tokens=65536, max_rows=4096, packed=True):
        flat, off, Lmax = model.pack_rows(id_rows, rows)
        batches.append((torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax))
    log(f"encoder leg: {n_chunks} chunks in {len(batches)} length-bucketed packed batches, weights resident")
    for ids, off, Lmax in batches[:3]:
        model.forward_packed(ids, off, Lmax)
    torch.cuda.synchronize()
    log("encoder warm-up done")
    if dist is not None:
        dist.barrier()
    # CODERAG_BENCH_EMBED_SYNC_EVERY=n (profiling runs only: tools/gpu_prof.sh sets it for the counter passes): wait for the
    # stream every n batches.  The leg queues ~27 000 launches without waiting; under `rocprofv3 --pmc` a queue that deep
    # crashed the profiled process inside the profiler's dispatch interception (profiles/README.md, "the --pmc SIGSEGV").
    sync_every = int(os.environ.get("CODERAG_BENCH_EMBED_SYNC_EVERY", "0"))
    if os.environ.get("CODERAG_BENCH_DUMP_MAPS"):          # load addresses of every library: what a crash's raw frames are symbolised against
        with open("/proc/self/maps") as f_in, open(os.environ["CODERAG_BENCH_DUMP_MAPS"], "w") as f_out:
            f_out.write("".join(ln for ln in f_in if ".so" in ln or "python" in ln))
    t0 = time.perf_counter()
    for bi, (ids, off, Lmax) in enumerate(batches):
        model.forward_packed(ids, off, Lmax)
        if sync_every and (bi + 1) % sync_every == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    log(f"encoder timed: {dt:.3f} s")
    flops = float(sum(drv.flops_per_chunk(int(n), cfg) for n in lengths))
    padded_tokens = int(sum(int(b[0].numel()) for b in batches))      # tokens the kernels processed (packed: the real ones)
    res = {"metric": "chunks embedded/s (UniXcoder-geometry bf16 HIP encoder)", "value": world * n_chunks / dt, "unit": "chunks/s",
           "chunks_per_gpu": int(n_chunks), "seconds": dt, "mean_tokens": float(lengths.mean()), "true_tokens": int(lengths.sum()),
           "padded_tokens": padded_tokens, "layout": "packed rows (no padding tokens)", "batches": len(batches),
           "dtype": "bf16 weights/activations, f32 accumulate/LN/softmax",
           "data": "synthetic ids + seeded random RoBERTa-base-geometry weights (no checkpoint offline)",
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, "algorithmic_flops": flops}}
    if rank == 0:
        res["ceiling"] = encoder_ceiling(np, torch, dev, [int(b[0].numel()) for b in batches], n_chunks, flops, dt, cfg)
    if cpu and rank == 0:
        from oracle import encoder as orc
        ocfg = orc.EncoderConfig()
        w = orc.random_weights(ocfg, 23)
        torch.set_num_threads(host_threads())
        t0 = time.perf_counter()
        done, toks = 0, 0
        for n in lengths:      # single-text calls, as the reference effectively issues them
            orc.forward(w, ocfg, orc.synthetic_ids(ocfg, [int(n)], 1))
            done += 1
            toks += int(n)
            if time.perf_counter() - t0 >= cpu_seconds:
                break
        cdt = time.perf_counter() - t0
        log(f"encoder CPU baseline done: {cdt:.1f} s")
        res["cpu_baseline"] = {"value": done / cdt, "unit": "chunks/s", "cores": host_threads(), "kind": "port",
                               "sample": f"oracle/encoder.py torch-fp32 forward, {done} single-text calls "
                                         f"(mean {toks / max(1, done):.0f} tokens), {cdt:.1f} s"}
    return res


def synth_chunk_ids(np, cfg, lengths, rng):
    """Token-id rows shaped like UniXcoder.tokenize output (unixcoder_provider.py:108-122): [<s>, <encoder-only>, </s>] + body + [</s>],
    body uniform over the vocabulary above the specials (SURVEY.md section 8d, C2 inputs)."""
    return [np.concatenate([[0, 5, 2], rng.integers(16, cfg.vocab_size, int(n) - 4), [2]]).astype(np.int32) if n >= 4
            else np.asarray([0, 5, 2, 2][:int(n)], np.int32) for n in lengths]


def c2_leg(np, torch, ffi, local_rank, n_chunks, rank, B, K, parity_chunks, dist=None, world=1):
    """BASELINE configs[1] as the ONE pipeline it names: n synthetic code chunks (the embed leg's seed and length mix) -> packed
    HIP encoder -> device-to-device crh_index_append -> batch-64 exact top-100 over the EMBEDDED vectors (the reference's
    embeddings/indexer.py:66-85 + query/vector_search.py:60-116 with the host round trips taken out).  Encoder outputs are
    anisotropic -- a large common component, tightly packed cosines -- which is where the scan's threshold / margin /
    candidate-buffer logic is least like the Gaussian corpus of the headline: `search_stats` reports what it did there, and
    ids + f32 score bits are compared with oracle/search on those same vectors (both stores).  Then, for a subsample and BOTH
    weight statistics (the deliberately sharp ones and HF-init ones), the whole GPU pipeline (bf16 encoder, bf16 store)
    against the fp32 pipeline (oracle encoder in fp32 + oracle f32 search): recall@100 / recall@10 / top-1."""
    from coderag_amd import encoder as drv
    from oracle import encoder as oenc
    from oracle import search as osr
    cfg = drv.EncoderConfig()
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rng = np.random.default_rng(1234 + rank)
    lengths = np.clip(np.round(np.exp(rng.normal(np.log(160.0), 0.8, n_chunks))), 8, 512).astype(np.int64)
    id_rows = synth_chunk_ids(np, cfg, lengths, rng)
    qrng = np.random.default_rng(4321)
    qlens = np.clip(np.round(np.exp(qrng.normal(np.log(40.0), 0.6, B))), 8, 512).astype(np.int64)      # queries are shorter than chunks
    q_rows = synth_chunk_ids(np, cfg, qlens, qrng)
    for i in range(min(B // 2, n_chunks)):            # half the queries are a corpus chunk with a tenth of its tokens replaced: a near neighbour exists
        row = id_rows[i].copy()
        if len(row) > 8:
            swap = qrng.choice(np.arange(3, len(row) - 1), max(1, (len(row) - 4) // 10), replace=False)
            row[swap] = qrng.integers(16, cfg.vocab_size, len(swap))
        q_rows[i] = row

    def build(init):
        return drv.HipUniXcoder(drv.synthetic_weights(cfg, 23, init=init), cfg, drv.HashTokenizer(cfg.vocab_size), local_rank)
    model = build("sharp")
    plan = model.plan_batches(lengths, max_tokens=65536, max_rows=4096, packed=True)
    batches = []
    for rows, _ in plan:
        flat, off, Lmax = model.pack_rows(id_rows, rows)
        batches.append((torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), Lmax, len(rows)))
    order = np.concatenate([np.asarray(rows, np.int64) for rows, _ in plan])        # index row r holds chunk order[r]
    codes = torch.zeros((max(b[3] for b in batches), 1), dtype=torch.int32, device=dev)
    res = {"workload": f"{n_chunks} synthetic code chunks (mean {lengths.mean():.0f} tokens) -> packed HIP encoder -> device-to-device "
                       f"crh_index_append -> batch-{B} exact top-{K} over the {n_chunks} embedded vectors",
           "data": "synthetic ids + seeded random RoBERTa-base-geometry weights (no checkpoint offline); queries: 32 corpus chunks with 10 % of "
                   "their tokens replaced + 32 fresh rows"}
    stores = {}
    multi = dist is not None and world > 1
    if multi:
        # N > 1 (embed row): every rank embeds ITS OWN n chunks (seed 1234 + rank) straight into its local shard -- no
        # collective, no vector crosses ranks -- the 64 queries are the same on every rank, each rank's exact local top-k (global
        # rows rank * n + local) meets the others in ONE all-gather and is merged on every rank: the sharded top-100 over world * n
        # embedded chunks.  Parity: every rank's local list against the oracle on ITS vectors, the merged list against a sort of
        # the gathered lists (the exact top-k of a union is the merge of the exact per-shard top-k).
        idx = ffi.Index(768, ffi.DTYPE_BF16, capacity_rows=n_chunks, n_code_cols=1, device=local_rank)
        for ids, off, Lmax, nr in batches[:2]:
            model.forward_packed(ids, off, Lmax)
        torch.cuda.synchronize()
        dist.barrier()
        outs = []
        t0 = time.perf_counter()
        for ids, off, Lmax, nr in batches:
            out = model.forward_packed(ids, off, Lmax)
            idx.append(out, codes=codes[:nr], stream=stream)
            outs.append(out)
        torch.cuda.synchronize()
        dist.barrier()
        t_index = time.perf_counter() - t0
        tt = torch.tensor([t_index], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_index = float(tt.item())
        qd = model.embed_ids([r.tolist() for r in q_rows])
        local, loc_s, loc_r, gathered, gat_s, gat_r = ffi.topk_exchange_buffers(torch, world, B, K, dev)
        mer_s = torch.empty((B, K), dtype=torch.float32, device=dev)
        mer_r = torch.empty((B, K), dtype=torch.int64, device=dev)

        def once():
            idx.search(qd, K, row_base=rank * n_chunks, out_scores=loc_s, out_rows=loc_r, stream=stream)
            dist.all_gather_into_tensor(gathered.view(-1), local)
            ffi.merge_topk(gat_s, gat_r, mer_s, mer_r, stream)
        for _ in range(3):
            once()
        idx.search_finish(stream)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            once()
        idx.search_finish(stream)
        torch.cuda.synchronize()
        dist.barrier()
        t_search = (time.perf_counter() - t0) / 20
        vecs = torch.cat(outs).cpu().numpy()
        es, er = osr.cosine_search(vecs, qd.cpu().numpy(), K, bf16=True)
        local_ok = bool(np.array_equal(loc_r.cpu().numpy() - rank * n_chunks, er) and np.array_equal(loc_s.cpu().numpy().view(np.uint32), es.view(np.uint32)))
        cat_s, cat_r = gat_s.permute(1, 0, 2).reshape(B, -1), gat_r.permute(1, 0, 2).reshape(B, -1)
        o1 = torch.argsort(cat_r, dim=1, stable=True)
        o2 = torch.argsort(torch.gather(cat_s, 1, o1), dim=1, descending=True, stable=True)
        order = torch.gather(o1, 1, o2)[:, :K]
        merged_ok = bool(torch.equal(torch.gather(cat_r, 1, order), mer_r) and torch.equal(torch.gather(cat_s, 1, order), mer_s))
        flag = torch.tensor([int(local_ok and merged_ok)], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        owners = np.bincount((mer_r.cpu().numpy().reshape(-1) // n_chunks).astype(np.int64), minlength=world).tolist()
        stats = idx.stats()
        idx.close()
        if rank != 0:
            return None
        res.update({"workload": f"{world} ranks x {n_chunks} synthetic code chunks: every rank embeds its own chunks into its local shard (no collective) -> "
                                f"batch-{B} exact top-{K} over the {world * n_chunks} embedded vectors (per-rank scan, one all-gather, merge)",
                    "value": world * n_chunks / t_index, "unit": "chunks/s (embed + index, all ranks)", "seconds_embed_plus_index": t_index,
                    "search": {"ms_per_batch": t_search * 1e3, "queries_per_s": B / t_search}, "search_stats_rank0": stats,
                    "merged_hits_by_owning_rank": owners,
                    "parity": {"every_rank_local_topk_bit_exact_vs_oracle_on_its_vectors_and_merge_equals_sorted_concat": bool(int(flag.item()) == 1),
                               "what": "rank-local ids and f32 score bits vs oracle/search on the rank's own embedded vectors; merged list vs a sort of the all-gathered lists; "
                                       "the flags of all ranks meet in one all-reduce(MIN)"}})
        log(f"c2 (x{world}): {res['value']:.0f} chunks/s embed+index over all ranks, sharded search {t_search * 1e3:.3f} ms/batch, parity {int(flag.item()) == 1}")
        return res
    for name, dtype in (("bf16", ffi.DTYPE_BF16), ("f32", ffi.DTYPE_F32)):
        idx = ffi.Index(768, dtype, capacity_rows=n_chunks, n_code_cols=1, device=local_rank)
        for ids, off, Lmax, nr in batches[:2]:
            model.forward_packed(ids, off, Lmax)
        torch.cuda.synchronize()
        outs = []
        t0 = time.perf_counter()
        for ids, off, Lmax, nr in batches:
            out = model.forward_packed(ids, off, Lmax)
            idx.append(out, codes=codes[:nr], stream=stream)           # device to device, on the forward's stream
            outs.append(out)
        torch.cuda.synchronize()
        t_index = time.perf_counter() - t0
        qd = model.embed_ids([r.tolist() for r in q_rows])
        r = timed_search(torch, idx, qd, K, None, 20, 3, stream)
        vecs = torch.cat(outs).cpu().numpy()
        qh = qd.cpu().numpy()
        es, er = osr.cosine_search(vecs, qh, K, bf16=(name == "bf16"))
        gs, gr = r["scores"].cpu().numpy(), r["rows"].cpu().numpy()
        stores[name] = {"ids_bit_exact": bool(np.array_equal(gr, er)), "scores_bit_exact": bool(np.array_equal(gs.view(np.uint32), es.view(np.uint32))),
                        "search_ms_per_batch": r["ms_per_step"], "search_stats": r["stats"], "scan_ms": r["scan_ms"],
                        "chunks_per_s_embed_plus_index": n_chunks / t_index}
        if name == "bf16":
            # how unlike the Gaussian corpus these vectors are: the common component and the spread of the top-100 scores
            vn = vecs / np.linalg.norm(vecs, axis=1, keepdims=True)
            mean_dir = vn.mean(0)
            res.update({"value": n_chunks / t_index, "unit": "chunks/s (embed + index, device to device)", "seconds_embed_plus_index": t_index,
                        "search": {"ms_per_batch": r["ms_per_step"], "queries_per_s": B / (r["ms_per_step"] * 1e-3), "step_ms_device": r["step_ms_device"]},
                        "search_stats": r["stats"],
                        "embedding_geometry": {"norm_of_mean_unit_vector": float(np.linalg.norm(mean_dir)),
                                               "top1_score_median": float(np.median(gs[:, 0])), "top100_score_median": float(np.median(gs[:, K - 1])),
                                               "top1_minus_top100_median": float(np.median(gs[:, 0] - gs[:, K - 1]))}})
            res["perturbed_chunk_queries_find_their_chunk_top1"] = float(np.mean([order[gr[i, 0]] == i for i in range(min(B // 2, n_chunks))]))
        idx.close()
        del outs, vecs
    res["parity"] = {"what": "ids and f32 score bits of the HIP search vs oracle/search on the SAME embedded vectors, all rows", **{f"{k}_store": v for k, v in stores.items()}}
    log(f"c2: {res['value']:.0f} chunks/s embed+index, search {res['search']['ms_per_batch']:.3f} ms/batch, "
        f"bit-exact bf16 {stores['bf16']['ids_bit_exact'] and stores['bf16']['scores_bit_exact']} f32 {stores['f32']['ids_bit_exact'] and stores['f32']['scores_bit_exact']}")

    # ---- end to end: GPU pipeline (bf16 encoder -> bf16 store -> HIP top-k) vs the fp32 pipeline (oracle encoder fp32 -> oracle f32 search)
    m = int(min(parity_chunks, n_chunks))
    e2e = {}
    if m > 0:
        sub = [id_rows[i] for i in range(m)]
        ocfg = oenc.EncoderConfig()
        kk = min(K, m)

        def oracle_embed(W, rows):      # fp32 torch graph of oracle/encoder.py evaluated on the GPU (device="cuda": see its docstring)
            out = np.empty((len(rows), 768), np.float32)
            idxs = np.argsort([len(r) for r in rows], kind="stable")
            for b0 in range(0, len(rows), 32):
                part = idxs[b0:b0 + 32]
                L = max(len(rows[i]) for i in part)
                ids = np.full((len(part), L), ocfg.pad_token_id, np.int64)
                for r_, i in enumerate(part):
                    ids[r_, : len(rows[i])] = rows[i]
                out[part] = oenc.forward(W, ocfg, ids, device=str(dev))
            return out
        def rec(a, b, n):
            return float(np.mean([len(set(x[:n].tolist()) & set(y[:n].tolist())) / n for x, y in zip(a, b)]))

        def gpu_side(mdl):
            g_c = mdl.embed_ids([r.tolist() for r in sub])
            g_q = mdl.embed_ids([r.tolist() for r in q_rows])
            idx = ffi.Index(768, ffi.DTYPE_BF16, capacity_rows=m, device=local_rank)
            idx.append(g_c, stream=stream)
            gs, gr = idx.search(g_q.cpu().numpy(), kk)
            idx.close()
            return g_c, gs, gr
        # three weight statistics (the published checkpoint cannot be loaded offline, so the regime it sits in is bracketed): the
        # deliberately sharp fixture, HF-init matrices with the sharp fixture's biases and LayerNorm parameters, plain HF init
        for label, init in (("sharp", "sharp"), ("hf_ln", "hf_ln"), ("hfinit", "hf")):
            mdl = model if init == "sharp" else build(init)
            W = oenc.to_device(oenc.random_weights(ocfg, 23, init=init), str(dev))
            g_c, gs, gr = gpu_side(mdl)
            o_c, o_q = oracle_embed(W, sub), oracle_embed(W, q_rows)
            ors, orr = osr.cosine_search(o_c, o_q, kk, bf16=False)
            # the opt-in levers (EncoderConfig): the residual stream in f32 through the LayerNorm kernels (12 bytes per element there
            # instead of 6), and the LayerNorm folded into the GEMMs (one rounding of a normalised activation less per LayerNorm,
            # profiles/r05_ln_fold_ab.md) -- what each buys against the fp32 pipeline, per weight statistic
            levers = {}
            for lever in ("residual_f32", "ln_fold"):
                fcfg = drv.EncoderConfig(**{lever: True})
                fmdl = drv.HipUniXcoder(drv.synthetic_weights(fcfg, 23, init=init), fcfg, drv.HashTokenizer(fcfg.vocab_size), local_rank)
                f_c, fs, fr = gpu_side(fmdl)
                fc = f_c.cpu().numpy()
                fcos = np.sum(fc * o_c, 1) / (np.linalg.norm(fc, axis=1) * np.linalg.norm(o_c, axis=1))
                levers[lever] = {f"recall_at_{kk}": rec(fr, orr, kk), "recall_at_10": rec(fr, orr, min(10, kk)),
                                 "top1_agreement": float(np.mean(fr[:, 0] == orr[:, 0])), "min_cosine_gpu_vs_fp32": float(fcos.min()),
                                 "mean_cosine_gpu_vs_fp32": float(fcos.mean()), "max_abs_score_difference_by_rank": float(np.max(np.abs(fs - ors)))}
                del fmdl
            gc = g_c.cpu().numpy()
            cos = np.sum(gc * o_c, 1) / (np.linalg.norm(gc, axis=1) * np.linalg.norm(o_c, axis=1))
            # north_star's bf16 criterion is a SCORE criterion ("within 1e-3 cosine score in bf16"): encoder outputs pack their
            # cosines tightly (the top-100 of a query span a few 1e-2), so ids near the cut-off swap on differences far below
            # it.  A returned id counts as right-within-eps when the fp32 pipeline scores it within eps of ITS k-th score.
            ref_all = osr.preprocess(o_q) @ osr.preprocess(o_c).T
            got_ref_scores = np.take_along_axis(ref_all, np.maximum(gr, 0), axis=1)
            within = {eps: float(np.mean(got_ref_scores >= ors[:, kk - 1:kk] - eps)) for eps in (1e-3, 1e-4)}
            by_rank = float(np.max(np.abs(gs - ors)))
            e2e[label] = {"chunks": m, "opt_in_levers": levers, "queries": int(len(q_rows)), f"recall_at_{kk}": rec(gr, orr, kk), "recall_at_10": rec(gr, orr, min(10, kk)),
                          "top1_agreement": float(np.mean(gr[:, 0] == orr[:, 0])), "min_cosine_gpu_vs_fp32": float(cos.min()),
                          "mean_cosine_gpu_vs_fp32": float(cos.mean()),
                          f"recall_at_{kk}_within_1e-3_of_the_fp32_kth_score": within[1e-3], f"recall_at_{kk}_within_1e-4": within[1e-4],
                          "max_abs_score_difference_by_rank": by_rank,
                          "fp32_top1_minus_topk_score_median": float(np.median(ors[:, 0] - ors[:, kk - 1]))}
            log(f"c2 e2e [{label}]: recall@{kk} {e2e[label][f'recall_at_{kk}']:.4f} (within 1e-3: {within[1e-3]:.4f}), top-1 {e2e[label]['top1_agreement']:.3f}, "
                f"min cos {cos.min():.6f}, score diff {by_rank:.1e}; " + "; ".join(
                    f"{k}: {v[f'recall_at_{kk}']:.4f} / {v['min_cosine_gpu_vs_fp32']:.6f} / {v['max_abs_score_difference_by_rank']:.1e}" for k, v in levers.items()))
            del W, mdl
            torch.cuda.empty_cache()
    res["end_to_end_vs_fp32_pipeline"] = dict(e2e, what="GPU: bf16 HIP encoder -> bf16 HIP store -> crh_search; reference side: oracle/encoder.py in fp32 "
                                              "(torch fp32 on the GPU for speed; pinned to its CPU evaluation by tests/test_c2_gpu.py) -> oracle f32 cosine search")
    return res


def encoder_ceiling(np, torch, dev, batch_tokens, n_chunks, flops, dt, cfg):
    """A practical bound for the encoder leg beside the 2.5 PFLOP/s of the roofline: per layer, the four GEMMs at the rate the
    vendor library (hipBLASLt through torch.nn.functional.linear, bf16, bias only -- no GELU, no residual) reaches on the same
    shapes ON THIS DEVICE, plus attention and the two LayerNorms at their HBM bound (every byte they must move, once, at the
    8 TB/s of the search roofline).  Measured here, outside every timed region; nothing of it is used by the product path."""
    H, F = cfg.hidden_size, cfg.intermediate_size
    T = int(np.median(batch_tokens))
    shapes = {"qkv": (3 * H, H), "oproj": (H, H), "ffn1": (F, H), "ffn2": (H, F)}
    g = torch.Generator(device=dev).manual_seed(3)
    us = {}
    for name, (N, K) in shapes.items():
        a = torch.randn((T, K), generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn((N,), generator=g, device=dev).to(torch.bfloat16)
        for _ in range(3):
            torch.nn.functional.linear(a, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.nn.functional.linear(a, w, b)
        e1.record()
        torch.cuda.synchronize()
        us[name] = e0.elapsed_time(e1) / 10 * 1e3
        del a, w, b
    gemm_s_per_token = sum(us.values()) * 1e-6 / T
    attn_bytes = (3 * H + H) * 2                      # per token and layer: QKV read once, context written once
    ln_bytes = (3 + 2) * H * 2                        # LayerNorm(x + residual): two reads + a write; LayerNorm: a read + a write
    hbm_s_per_token = (attn_bytes + ln_bytes) / (HBM_PEAK_GBS * 1e9)
    total_tokens = float(sum(batch_tokens))
    ceil_s = cfg.num_layers * total_tokens * (gemm_s_per_token + hbm_s_per_token)
    return {"seconds": ceil_s, "chunks_per_s": n_chunks / ceil_s, "tflops": flops / ceil_s / 1e12, "frac_of_mfma_peak": flops / ceil_s / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "achieved_frac_of_ceiling": ceil_s / dt, "vendor_gemm_us_at_T": dict(us, T=T),
            "vendor_gemm_tflops": {k: 2.0 * T * shapes[k][0] * shapes[k][1] / (v * 1e-6) / 1e12 for k, v in us.items()},
            "hbm_bound_us_per_layer_at_T": {"attention": attn_bytes * T / (HBM_PEAK_GBS * 1e3), "layernorms": ln_bytes * T / (HBM_PEAK_GBS * 1e3)},
            "what": "sum over layers of [4 GEMMs at the vendor library's measured rate on these shapes (bias only) + attention and LayerNorm "
                    "bytes at 8 TB/s]; embedding gather, pool and launch gaps count as zero"}


def _source_files():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "**", "*.py"), recursive=True)
                   + glob.glob(os.path.join(ROOT, "code-rag_amd", "csrc", "*"))
                   + glob.glob(os.path.join(ROOT, "code-rag_amd", "csrc_host", "*.cpp")))
    return [f for f in files if "gpurun_out" not in f and os.path.getsize(f) > 0]


def synth_checkpoint(np, torch, box: dict) -> str:
    """A LOCAL UniXcoder-shaped checkpoint directory (no hub, no network): seeded 12-layer RoBERTa-base-geometry weights +
    a byte-level BPE vocabulary trained here on this repo's sources (8000 entries incl. <encoder-only>), written as
    config.json / model.safetensors / vocab.json / merges.txt -- exactly what `HipUniXcoderProvider(model=<dir>)` loads."""
    if "dir" in box:
        return box["dir"]
    import tempfile
    from safetensors.torch import save_file
    from tokenizers import ByteLevelBPETokenizer
    from coderag_amd import encoder as drv
    d = tempfile.mkdtemp(prefix="coderag_ckpt_")
    tr = ByteLevelBPETokenizer(add_prefix_space=False)
    tr.train(_source_files(), vocab_size=8000, min_frequency=2, show_progress=False,
             special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>", "<encoder-only>"])
    tr.save_model(d)
    cfg = drv.EncoderConfig(vocab_size=8000)
    json.dump({"vocab_size": 8000, "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 3072,
               "max_position_embeddings": 1026, "type_vocab_size": 10}, open(os.path.join(d, "config.json"), "w"))
    weights = drv.synthetic_weights(cfg, 31)
    save_file({k: torch.from_numpy(v) for k, v in weights.items()}, os.path.join(d, "model.safetensors"))
    box.update(dir=d, cfg=cfg, weights=weights)
    return d


def embed_e2e_leg(np, torch, local_rank, n_texts, box):
    """The surface the reference calls (providers/unixcoder_provider.py:195-215): texts -> `embed_batch` -> list[list[float]],
    through the native byte-level BPE tokenizer, the length-bucketed HIP forward, the D2H copy and `.tolist()`.
    (i) one `embed_batch` call over all texts; (ii) the reference's call shape: 32 coroutines, each `embed_batch(~600 texts,
    batch_size=100)` (embeddings/embedder.py:58-63), coalesced by the provider's dynamic batching."""
    import asyncio
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    d = synth_checkpoint(np, torch, box)
    os.environ["CODERAG_HIP_DEVICE"] = str(local_rank)
    texts = []
    for f in _source_files():
        s = open(f, encoding="utf-8", errors="ignore").read()
        texts += [s[i:i + 700] for i in range(0, len(s), 700)]
    rng = np.random.default_rng(0)
    texts = [texts[i] for i in rng.integers(0, len(texts), n_texts)]
    prov = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model=d))
    box["provider"] = prov

    async def one_call():
        return await prov.embed_batch(texts, batch_size=len(texts))

    async def reference_shape():
        prov.set_concurrency(64)
        per = (len(texts) + 31) // 32
        parts = await asyncio.gather(*(prov.embed_batch(texts[i:i + per], batch_size=100) for i in range(0, len(texts), per)))
        return [v for p in parts for v in p]

    asyncio.run(prov.embed_batch(texts[:2000], batch_size=2000))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vecs = asyncio.run(one_call())
    t_one = time.perf_counter() - t0
    sub0 = prov.submissions
    t0 = time.perf_counter()
    vecs2 = asyncio.run(reference_shape())
    t_ref = time.perf_counter() - t0
    assert len(vecs) == len(vecs2) == n_texts and isinstance(vecs[0], list) and isinstance(vecs[0][0], float) and len(vecs[0]) == 768
    model = prov._load()
    ids, lens = model.tok.encode_bodies(texts[:4000], 508)
    mean_tok = float(np.minimum(lens, 508).mean() + 4)
    from coderag_amd import encoder as drv
    flops = float(sum(drv.flops_per_chunk(int(min(n, 508)) + 4, box["cfg"]) for n in lens)) * (n_texts / len(lens))
    # the two call shapes run different batch compositions: same text -> same vector to bf16 accuracy
    a, b = np.asarray(vecs[:512], np.float32), np.asarray(vecs2[:512], np.float32)
    cos = float(np.min(np.sum(a * b, 1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))))
    log(f"embed_e2e: {n_texts / t_one:.0f} texts/s (one call), {n_texts / t_ref:.0f} texts/s (reference call shape)")
    return {"metric": "texts embedded/s through EmbeddingProvider.embed_batch -> python float lists", "value": n_texts / t_one,
            "unit": "texts/s", "texts": n_texts, "mean_tokens": mean_tok, "seconds": t_one,
            "reference_call_shape": {"value": n_texts / t_ref, "seconds": t_ref, "gpu_submissions": prov.submissions - sub0,
                                     "what": "32 concurrent embed_batch(~600 texts, batch_size=100) coroutines, coalesced"},
            "data": "700-character slices of this repo's sources; byte-level BPE vocabulary (8000) trained on them; seeded weights",
            "tokenizer": "native C++ byte-level BPE (lib/libcoderag_tok.so)",
            "roofline": {"bound": "mfma", "achieved": flops / t_one / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / t_one / 1e12 / MFMA_BF16_PEAK_TFLOPS, "algorithmic_flops": flops,
                         "note": "end to end: tokenizer, H2D, forward, D2H and .tolist() are all inside the time"},
            "parity": {"min_cosine_same_text_across_batch_shapes": cos}}


def index_e2e_leg(np, torch, local_rank, box, n_files=600):
    """Indexing END TO END through the reference-shaped surfaces only (src/lattice/embeddings/indexer.py:46-119,
    pipeline/orchestrator.py:652-656): parsed files -> CodeChunker -> Embedder -> HipUniXcoderProvider (native tokenizer, HIP
    encoder) -> HipVectorStore.upsert.  `sequential`: VectorIndexer.index_files as the reference runs it (per file: update
    check, delete, chunk, embed_with_progress, upsert).  `value`: VectorIndexer.index_files_batched -- the same outcome in one
    pass (one update check, one delete job, all files chunked, ONE coalesced embedding submission handed over as an array, ONE
    upsert).  Both runs are checked against each other: same chunk count, and a stored chunk's text finds that chunk."""
    import asyncio
    import types
    from pathlib import Path
    from coderag_amd.embedder import Embedder
    from coderag_amd.indexer import CodeChunker, VectorIndexer
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.store import HipVectorStore
    d = synth_checkpoint(np, torch, box)
    os.environ["CODERAG_HIP_DEVICE"] = str(local_rank)
    lines = "\n".join(open(f, encoding="utf-8", errors="ignore").read() for f in _source_files() if f.endswith(".py")).split("\n")
    rng = np.random.default_rng(0)

    def parsed_file(i):
        ents = []
        for j in range(int(rng.integers(8, 40))):
            a = int(rng.integers(0, len(lines) - 40))
            code = "\n".join(lines[a:a + int(rng.integers(6, 40))])
            ents.append(types.SimpleNamespace(type=types.SimpleNamespace(value="function"), name=f"fn_{i}_{j}", qualified_name=f"mod{i}.fn_{i}_{j}",
                                              signature=f"def fn_{i}_{j}(x)", docstring="Does things.", code=code, start_line=10 * j + 1, end_line=10 * j + 9))
        info = types.SimpleNamespace(path=Path(f"/proj/mod{i}.py"), content_hash=f"h{i}", language=types.SimpleNamespace(value="python"))
        return types.SimpleNamespace(file_info=info, content="", all_entities=ents)
    files = [parsed_file(i) for i in range(n_files)]
    provider = box.get("provider") or HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model=d))
    embedder = Embedder(provider_instance=provider)
    probe = files[7].all_entities[3]
    probe_text = "\n".join([probe.signature, f'"""{probe.docstring}"""', probe.code])

    async def one(batched):
        async with HipVectorStore(device=local_rank, dim=768, dtype="bf16", initial_capacity=1 << 16) as store:
            await store.create_collections()
            indexer = VectorIndexer(store, embedder, CodeChunker(max_tokens=1000, overlap_tokens=200))
            t0 = time.perf_counter()
            n = await (indexer.index_files_batched if batched else indexer.index_files)(files, project_name="proj")
            dt = time.perf_counter() - t0
            qv = await embedder.embed(probe_text)
            top = (await store.search("code_chunks", qv, limit=1))[0]
            again = await (indexer.index_files_batched if batched else indexer.index_files)(files[:50], project_name="proj")      # unchanged: skipped
            return n, dt, (top["payload"]["entity_name"], top["payload"]["file_path"]), again, (await store.get_collection_info("code_chunks")).points_count
    asyncio.run(embedder.embed_batch(["warm up"] * 64))
    # the bound of this leg: the encoder alone on the very texts the files chunk into (they are longer than the embed leg's mix)
    all_texts = [c.content for f in files for c in CodeChunker(max_tokens=1000, overlap_tokens=200).chunk_file(f, project_name="proj")]
    asyncio.run(embedder.embed_array(all_texts[:4096]))
    t0 = time.perf_counter()
    asyncio.run(embedder.embed_array(all_texts))
    t_enc = time.perf_counter() - t0
    mean_tok = float(np.minimum(provider._load().tok.encode_bodies(all_texts[:4000], 508)[1], 508).mean() + 4)
    n_b, t_b, top_b, again_b, count_b = asyncio.run(one(True))
    n_s, t_s, top_s, again_s, count_s = asyncio.run(one(False))
    same = bool(n_b == n_s == count_b == count_s and top_b == top_s == ("mod7.fn_7_3", "/proj/mod7.py") and again_b == again_s == 0)
    log(f"index_e2e: batched {n_b / t_b:.0f} chunks/s (the encoder alone on these texts, mean {mean_tok:.0f} tokens: {len(all_texts) / t_enc:.0f}), "
        f"sequential (the reference's flow) {n_s / t_s:.0f} chunks/s, same outcome {same}")
    return {"metric": "chunks indexed/s through VectorIndexer -> CodeChunker -> Embedder -> provider -> HipVectorStore.upsert", "value": n_b / t_b,
            "unit": "chunks/s", "files": n_files, "chunks": n_b, "seconds": t_b, "mean_tokens": mean_tok,
            "encoder_alone": {"value": len(all_texts) / t_enc, "seconds": t_enc, "what": "ONE embed_array call over the same chunk texts (tokenizer + packed forward + copy back), "
                              "nothing else: the bound of this leg", "leg_over_encoder_alone": (n_b / t_b) / (len(all_texts) / t_enc)},
            "flow": "VectorIndexer.index_files_batched: one update check, one delete job, all files chunked, one coalesced embedding submission (float32 array), one upsert",
            "sequential": {"value": n_s / t_s, "seconds": t_s, "flow": "VectorIndexer.index_files (indexer.py:96-119): per file update check, delete, chunk, embed_with_progress, upsert"},
            "data": "600 synthetic parsed files of 8-40 entities cut from this repo's Python sources; byte-level BPE vocabulary (8000) trained on them; seeded 12-layer weights",
            "parity": {"batched_equals_sequential": same, "what": "chunk counts, collection size, a stored chunk's own text finds that chunk first, unchanged files are skipped on a second run"}}


def code_chunks(limit: int = 1000):
    """~1k code chunks of this repo: one per top-level / class-level def of every Python file, 40-line windows of the HIP
    and C++ sources -- the stand-in for `tests/fixtures` of the reference (BASELINE configs[0]), which does not travel."""
    import ast
    chunks = []
    for f in _source_files():
        src = open(f, encoding="utf-8", errors="ignore").read()
        rel = os.path.relpath(f, ROOT)
        lines = src.splitlines()
        if f.endswith(".py"):
            try:
                tree = ast.parse(src)
            except SyntaxError:
                continue
            for node in ast.walk(tree):
                if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) and node.end_lineno - node.lineno >= 2:
                    chunks.append((rel, node.name, node.lineno, "\n".join(lines[node.lineno - 1:node.end_lineno])[:4000]))
            for i in range(0, len(lines), 60):          # + module-level windows (imports, constants, scripts)
                chunks.append((rel, f"{os.path.basename(f)}:{i + 1}", i + 1, "\n".join(lines[i:i + 60])))
        else:
            for i in range(0, len(lines), 40):
                chunks.append((rel, f"{os.path.basename(f)}:{i + 1}", i + 1, "\n".join(lines[i:i + 40])))
    chunks.sort(key=lambda c: (c[0], c[2], c[1]))
    step = max(1, len(chunks) // limit)
    return chunks[::step][:limit]


def c1_leg(np, torch, local_rank, box, cpu_seconds):
    """BASELINE configs[0] shape -- ~1k code chunks embedded, then cosine top-10 ONE QUERY PER CALL -- through the
    reference-shaped surfaces on the GPU (HipUniXcoderProvider -> Embedder -> HipVectorStore.upsert -> VectorSearcher.
    search_code: query/vector_search.py:60-116), and the same texts through the CPU oracles (torch-fp32 encoder single-text
    calls as providers/unixcoder_provider.py:176-193 issues them, scalar cosine scan) as `cpu_baseline` (time-boxed sample)."""
    import asyncio
    from coderag_amd.embedder import Embedder
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.store import CollectionName, HipVectorStore
    from coderag_amd.vector_search import VectorSearcher
    d = synth_checkpoint(np, torch, box)
    os.environ["CODERAG_HIP_DEVICE"] = str(local_rank)
    chunks = code_chunks(1000)
    texts = [c[3] for c in chunks]
    qidx = np.random.default_rng(5).choice(len(chunks), 64, replace=False)
    queries = [" ".join(chunks[i][3].split()[:24]) or chunks[i][1] for i in qidx]   # the head of a chunk as a natural-language-ish query
    prov = box.get("provider") or HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model=d))

    async def gpu_side():
        emb = Embedder(provider_instance=prov, max_concurrent=5)
        store = HipVectorStore(device=local_rank, dim=768, dtype="f32")
        await store.connect()
        await store.create_collections()
        t0 = time.perf_counter()
        vecs = await emb.embed_with_progress(texts)
        t_embed = time.perf_counter() - t0
        payloads = [{"file_path": c[0], "entity_type": "function", "entity_name": c[1], "language": "python", "start_line": c[2],
                     "end_line": c[2], "content": c[3], "graph_node_id": None, "content_hash": "", "project_name": "bench"} for c in chunks]
        await store.upsert(CollectionName.CODE_CHUNKS.value, [str(i) for i in range(len(chunks))], vecs, payloads)
        searcher = VectorSearcher(store, emb)
        await searcher.search_code(queries[0], limit=10)
        t0 = time.perf_counter()
        hits = [await searcher.search_code(q, limit=10) for q in queries]
        t_query = time.perf_counter() - t0
        await store.close()
        return vecs, hits, t_embed, t_query

    vecs, hits, t_embed, t_query = asyncio.run(gpu_side())
    res = {"workload": f"{len(chunks)} code chunks of this repo (mean {np.mean([len(t) for t in texts]):.0f} chars) -> embed_with_progress -> "
                       "upsert -> 64 x search_code(limit=10), one query per call",
           "embed": {"value": len(chunks) / t_embed, "unit": "chunks/s", "seconds": t_embed},
           "query": {"value": len(queries) / t_query, "unit": "queries/s (embed query + top-10, one per call)", "seconds": t_query}}
    if cpu_seconds <= 0:
        return res
    # ---- the same path on the host cores: oracle encoder (fp32, single-text calls) + oracle scalar cosine scan
    from oracle import encoder as oenc
    from oracle import search as osr
    model = prov._load()
    ocfg = oenc.EncoderConfig(vocab_size=box["cfg"].vocab_size)
    torch.set_num_threads(host_threads())

    def cpu_embed(text):
        ids, lens = model.tok.encode_bodies([text], 508)
        b = int(min(lens[0], 508))
        row = np.concatenate([[model.tok.cls_id, model.tok.enc_only_id, model.tok.sep_id], ids[0, :b], [model.tok.sep_id]])
        return oenc.forward(box["weights"], ocfg, row[None, :])[0]
    t0 = time.perf_counter()
    cpu_q = [cpu_embed(q) for q in queries]
    t_q_embed = time.perf_counter() - t0
    cpu_vecs, done = [], 0
    order = list(dict.fromkeys([int(i) for i in qidx] + list(range(len(chunks)))))   # the queries' own chunks first
    t0 = time.perf_counter()
    for i in order:
        cpu_vecs.append(cpu_embed(texts[i]))
        done += 1
        if time.perf_counter() - t0 >= cpu_seconds:
            break
    t_c_embed = time.perf_counter() - t0
    X = np.stack(cpu_vecs)
    Xp = osr.preprocess(X)
    t0 = time.perf_counter()
    cpu_top = [osr.search(Xp, osr.preprocess(q[None, :]), 10)[1][0] for q in cpu_q]     # one query per call
    t_scan = time.perf_counter() - t0
    # parity of the pipeline: GPU embeddings of the same chunks vs the fp32 oracle; top-10 agreement on the embedded subset
    G = np.asarray([vecs[i] for i in order[:done]], np.float32)
    cos = np.sum(G * X, 1) / (np.linalg.norm(G, axis=1) * np.linalg.norm(X, axis=1))
    Gp = osr.preprocess(G)
    gq = np.asarray(asyncio.run(prov.embed_batch(queries, batch_size=64)), np.float32)
    gpu_top = osr.search(Gp, osr.preprocess(gq), 10)[1]
    rec = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / 10.0 for a, b in zip(gpu_top, cpu_top)]))
    top1 = float(np.mean([a[0] == b[0] for a, b in zip(gpu_top, cpu_top)]))
    res["cpu_baseline"] = {"value": done / t_c_embed, "unit": "chunks/s", "cores": host_threads(), "kind": "port",
                           "queries_per_s": len(queries) / (t_q_embed + t_scan),
                           "sample": f"oracle/encoder.py torch-fp32, {done} of the {len(chunks)} chunks single-text in {t_c_embed:.1f} s; "
                                     f"{len(queries)} queries: embed {t_q_embed:.1f} s + oracle scalar top-10 over {done} rows, one per call, {t_scan * 1e3:.0f} ms"}
    res["parity"] = {"chunks_compared": done, "min_cosine_gpu_vs_fp32_oracle": float(cos.min()), "mean_cosine_gpu_vs_fp32_oracle": float(cos.mean()),
                     "recall_at_10_gpu_embeddings_vs_oracle_embeddings": rec, "top1_agreement": top1,
                     "hits_per_query": float(np.mean([len(h) for h in hits]))}
    log(f"c1: GPU {len(chunks) / t_embed:.0f} chunks/s, CPU {done / t_c_embed:.1f} chunks/s, min cos {cos.min():.5f}, recall@10 {rec:.3f}")
    return res


def cpu_baseline(np, B, K, D, seconds):
    """Exact cosine top-k on the host cores with the oracle's BLAS scan (oracle/search.py:search_blas):
    the stand-in for "Qdrant exact scan" (BASELINE.md section 3).  Bounded sample: 1M rows, time-boxed."""
    from oracle import search as orc
    from threadpoolctl import threadpool_limits
    rows = 1_000_000
    log("search CPU baseline: generating 1M-row sample")
    rng = np.random.default_rng(20251226)
    x = rng.standard_normal((rows, D), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = orc.preprocess(np.random.default_rng(7).standard_normal((B, D)).astype(np.float32))
    with threadpool_limits(limits=host_threads()):
        orc.search_blas(x, q, K)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < seconds:
            orc.search_blas(x, q, K)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
    log("search CPU baseline done")
    return {"value": B / dt * (rows / 1e7), "unit": "queries/s per 10M-row shard (row.query pairs/s / 1e7)",
            "cores": host_threads(), "kind": "port",
            "sample": f"oracle BLAS exact scan (f32 sgemm + argpartition), {rows} rows x {B} queries top-{K}, "
                      f"{reps} batches of {dt * 1e3:.0f} ms, scaled to 10M rows"}


if __name__ == "__main__":
    main()
