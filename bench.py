#!/usr/bin/env python3
"""bench.py -- compress throughput of the x3 hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (K1 scan -> K2 parse -> K3 code) over the workload, inputs already resident
in HBM, outputs left in HBM (x3h_compress_chunks_dev).

N = 1 (default): BASELINE.json configs[1] -- ONE dickens-sized stream (10 192 446 bytes of synthetic English-like text; Silesia
  itself is not available offline), -w 64 -t 256, bit-exact x3 code stream: its sha256 is checked against the REAL reference's
  (tests/golden/manifest_sha.json) inside the run.  The library overlaps the stages of one long stream (api.hip run_pipelined).
  Secondary figures in the same JSON line: the same bytes as 64/128/256/512 independent chunks (the only way the path shards,
  SURVEY.md 8(e)), a 256 MiB batch of 1024 chunks, one GPU's share of config 4, the host-buffer (PCIe-inclusive) rate.
N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[3] in its weak form -- every rank codes 16
  chunks x 8 MiB of the Zipf byte stream (rank r: chunks 16r .. 16r+15 of the 128), then ONE RCCL gather of the finished streams
  to rank 0, which frames them as an X3C1 container.  No data-path collective.  `one_stream_per_rank` keeps the single-stream
  figure as a secondary key.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from x3_compressor_amd import _lib, container, synth
from x3_compressor_amd import dist as xdist

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
HBM_COPY_PEAK = 6.29e12  # B/s, the same guide's measured copy rate (SURVEY.md 8(d) quotes both)
REF = os.path.join(ROOT, "oracle", "_ref", "x3")
CHUNK4 = 8 << 20   # config 4: 128 chunks x 8 MiB, 16 per GPU


def run_reference(sample: bytes, w_kib: int, t: int):
    """The real reference (oracle/_ref/x3, built from /root/reference in the build container) on one sample, one core;
    falls back to the oracle port if the prebuilt binary is absent.  -> (seconds of its own 'elapsed time', stream, kind)"""
    with tempfile.TemporaryDirectory() as d:
        i, o = os.path.join(d, "in"), os.path.join(d, "out")
        open(i, "wb").write(sample)
        if os.path.exists(REF):
            r = subprocess.run([REF, "-z", "-f", "-w", str(w_kib), "-t", str(t), i, o], capture_output=True, text=True)
            if r.returncode == 0:
                sec = float([l for l in r.stderr.splitlines() if l.startswith("elapsed time:")][0].split(":")[1])
                return sec, open(o, "rb").read(), "reference"
        x3o = os.path.join(ROOT, "oracle", "x3o")
        if not os.path.exists(x3o):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "x3o"], check=True, capture_output=True)
        r = subprocess.run([x3o, "-z", "-w", str(w_kib), "-t", str(t), i, o], capture_output=True, text=True, check=True)
        sec = float([l for l in r.stderr.splitlines() if l.startswith("elapsed")][0].split()[1])
        return sec, open(o, "rb").read(), "port"


def run_reference_decode(stream: bytes):
    """the real reference's decoder (oracle/_ref/x3 -d, x3.c:613-645) on one stream, one core -> (seconds of its own 'elapsed time', bytes), or None"""
    if not os.path.exists(REF):
        return None
    with tempfile.TemporaryDirectory() as d:
        i, o = os.path.join(d, "in.x3"), os.path.join(d, "out")
        open(i, "wb").write(stream)
        r = subprocess.run([REF, "-d", "-f", i, o], capture_output=True, text=True)
        if r.returncode != 0:
            return None
        sec = float([l for l in r.stderr.splitlines() if l.startswith("elapsed time:")][0].split(":")[1])
        return sec, open(o, "rb").read()


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (PMC numbers come from separate profiling passes, tools/r03_refresh.sh)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return c[-1] if c else None


def timed(fn, sync):
    sync()
    t0 = time.perf_counter()
    r = fn()
    sync()
    return time.perf_counter() - t0, r


def chunk_batch(ctx, d_in, total, cb, prm, dev, reps=2):
    """one batch of independent chunks of `cb` bytes, device-resident in/out -> (seconds, lens, stats)"""
    off = np.array(list(range(0, total, cb)) + [total], dtype=np.uint64)
    stride = (cb + (cb >> 1) + 4096 + 3) & ~3
    d_out = torch.empty(stride * (len(off) - 1), dtype=torch.uint8, device=dev)
    best = None
    for it in range(reps + 1):  # the first run allocates the workspace
        dt, (lens, st) = timed(lambda: ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride), torch.cuda.synchronize)
        if it and (best is None or dt < best[0]):
            best = (dt, lens, st, off, d_out, stride)
    return best


PINNED4 = (0, 1, 15, 64, 127)  # chunks of config 4's 128 x 8 MiB Zipf stream whose reference streams are pinned (tests/golden/manifest_sha.json)


def pinned_chunk_ok(manifest, chunk, stream: bytes):
    """True / False: `stream` against the REAL reference's sha256 of chunk `chunk` of config 4 (x3 -z -w 64 -t 256 of that chunk alone,
    x3.c:372-434,593-611); None if that chunk is not pinned"""
    e = manifest.get(f"cfg4_zipf_chunk{chunk}_8m_w64_t256")
    if e is None:
        return None
    return bool(len(stream) == e["output_len"] and hashlib.sha256(stream).hexdigest() == e["output_sha256"])


def pinned_pieces_ok(manifest, base, stream_of, chunk_bytes=None, names=None):
    """the chunks of one of the timed many-stream batches whose streams the REAL reference wrote (tests/golden/make_golden_sha.py, `piece` entries of `base`):
    stream_of(chunk index) -> bytes of the TIMED call's output.  -> (sorted chunk indices checked, all equal?)"""
    checked, ok = [], True
    for n, e in sorted(manifest.items()):
        if e.get("generator") != "piece" or e["generator_args"]["base"] != base or (names is not None and n not in names):
            continue
        c = names[n] if names is not None else e["generator_args"]["start"] // chunk_bytes
        s = stream_of(c)
        good = len(s) == e["output_len"] and hashlib.sha256(s).hexdigest() == e["output_sha256"]
        if not good:
            print(f"bench.py: {n}: chunk {c} of the timed batch differs from the real reference's stream", file=sys.stderr)
        checked.append(c)
        ok = ok and good
    return sorted(checked), bool(checked) and ok


def emit_line(line):
    """the ONE JSON line of the contract, on the process's real stdout (main() points fd 1 at stderr meanwhile: RCCL prints a version banner to
    stdout when a communicator is made, and nothing but the line may appear there)"""
    os.write(_REAL_STDOUT, (json.dumps(line) + "\n").encode())


_REAL_STDOUT = 1


def main():
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=synth.DICKENS_BYTES)
    ap.add_argument("--w", type=int, default=64)
    ap.add_argument("--t", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=256 * 1024)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the timed workload (profiling runs)")
    ap.add_argument("--no-config35", action="store_true", help="skip the config 3 / config 5 legs (about a minute of input generation and GPU decode)")
    ap.add_argument("--many-chunks-mib", type=int, default=256, help="secondary: a batch of this many MiB cut into 256 KiB chunks (0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run (any world size)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    prm = _lib.make_params(w_kib=args.w, t=args.t)
    ctx = _lib.X3Context(local)
    default_workload = (args.bytes, args.w, args.t) == (synth.DICKENS_BYTES, 64, 256)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed workload ------------------------------------------------------------------------------------------------------
    # (X3_BENCH_LEG=config4 under torch.distributed.run with ONE rank rehearses the N > 1 leg on a single GPU)
    config4_leg = world > 1 or (distributed and os.environ.get("X3_BENCH_LEG") == "config4")
    if not config4_leg:
        # configs[1]: one dickens-sized stream
        data = synth.english_like(args.bytes)
        d_in = torch.from_numpy(data).to(dev)
        stride = (2 * args.bytes + 4096 + 3) & ~3
        d_out = torch.empty(stride, dtype=torch.uint8, device=dev)
        offsets = np.array([0, args.bytes], dtype=np.uint64)
        unit_bytes = args.bytes

        def step():
            return ctx.compress_chunks_dev(d_in.data_ptr(), offsets, prm, d_out.data_ptr(), stride)
    else:
        # configs[3], weak form: rank r codes chunks 16r .. 16r+15 of the Zipf stream, one gather, rank 0 frames the container
        per = 16
        data = synth.zipf_bytes(per * CHUNK4, offset=rank * per * CHUNK4)
        d_in = torch.from_numpy(data).to(dev)
        stride = (CHUNK4 + (CHUNK4 >> 2) + 4096 + 3) & ~3
        d_out = torch.empty(stride * per, dtype=torch.uint8, device=dev)
        offsets = np.arange(0, (per + 1) * CHUNK4, CHUNK4, dtype=np.uint64)
        unit_bytes = per * CHUNK4
        slot = xdist.default_slot_bytes(per * CHUNK4, per)
        container_bytes, keep_last, gather_s = [0], [None], [0.0]

        def step():
            lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), offsets, prm, d_out.data_ptr(), stride)
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            got = xdist.gather_device_streams(d_out, stride, lens, slot, per, to_host=False)  # the ONE exchange step of the path (RCCL over xGMI)
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - g0  # (this rank's view: pack + one gather; rank 0 also waits for the slowest sender)
            if rank == 0:  # final bitstream concat, left in HBM like every output: X3C1 frame + the ranks' payloads back to back
                all_lens, payloads = got
                head = container.header([CHUNK4] * len(all_lens), all_lens, prm)
                blob = torch.cat([torch.frombuffer(bytearray(head), dtype=torch.uint8).to(dev)] + payloads)
                container_bytes[0] = int(blob.numel())
                keep_last[0] = blob
            return lens, st

    for _ in range(args.warmup):
        step()
    barrier()
    if config4_leg:
        gather_s[0] = 0.0
    t0 = time.perf_counter()
    acc = {k: 0.0 for k in ("ms_scan", "ms_parse", "ms_code", "ms_features", "ms_modes", "ms_coder", "ms_emit", "ms_total")}
    for _ in range(args.steps):
        lens, st = step()
        for k in acc:
            acc[k] += getattr(st, k)
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt * 1e3 / args.steps
    value = unit_bytes * world / (dt / args.steps) / 1e6
    ms = {k: v / args.steps for k, v in acc.items()}
    out_len = int(np.asarray(lens).sum())

    line = {
        "metric": "compress MB/s + ratio, Silesia 'dickens' -w 64 -t 256, at 1/2/4/8 MI355X",
        "value": round(value, 3), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
    }

    if config4_leg:
        # secondary: the single-stream figure with one dickens-sized stream per rank (no exchange)
        sdata = synth.english_like(synth.DICKENS_BYTES, seed=0xD1C4E25 + rank)
        sd_in = torch.from_numpy(sdata).to(dev)
        sstride = (2 * sdata.size + 4096 + 3) & ~3
        sd_out = torch.empty(sstride, dtype=torch.uint8, device=dev)
        soff = np.array([0, sdata.size], dtype=np.uint64)
        ctx.compress_chunks_dev(sd_in.data_ptr(), soff, prm, sd_out.data_ptr(), sstride)
        barrier()
        t1 = time.perf_counter()
        ctx.compress_chunks_dev(sd_in.data_ptr(), soff, prm, sd_out.data_ptr(), sstride)
        barrier()
        sdt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
        if rank == 0:
            line["config"] = {"workload": f"config 4 (weak form): {per} chunks x 8 MiB of the Zipf(s=1) byte stream per GPU (rank r = chunks {per}r..{per}r+{per - 1} "
                                          f"of the 128), -w {args.w} -t {args.t}, every chunk its own bit-exact x3 stream, one RCCL gather to rank 0, X3C1 container",
                              "window_kib": args.w, "max_match_count": args.t, "chunks_per_gpu": per, "chunk_bytes": CHUNK4,
                              "parallelism": f"chunks over {world} GPUs, no data-path collective, one gather"}
            line["ratio"] = round(unit_bytes / out_len, 4)
            line["container_bytes"] = container_bytes[0]
            # the exchange step, so that the first real multi-GPU run validates itself: how many ranks took part in the ONE gather, and what it cost rank 0 per step
            line["rccl_ranks"] = int(dist.get_world_size())
            line["gather_backend"] = str(dist.get_backend())
            line["gather_ms"] = round(gather_s[0] * 1e3 / args.steps, 3)
            line["gather_frame_bytes_per_rank"] = int(8 * (3 + per) + slot)
            prm_echo, chunks = container.unpack(keep_last[0].cpu().numpy().tobytes())  # (outside the timed region) the container parses back
            line["container_ok"] = bool(prm_echo is not None and len(chunks) == per * world and all(r == CHUNK4 for r, _ in chunks))
            # bit-exactness of the timed form, asserted in the run: rank 0 holds every rank's streams -- those of the pinned chunks are compared
            # with the real reference's (chunk c lives on rank c // 16; a run with fewer ranks checks the pinned chunks it has)
            man4 = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest_sha.json")))
            default4 = (args.w, args.t) == (64, 256)
            checked = {c: pinned_chunk_ok(man4, c, chunks[c][1]) for c in PINNED4 if c < len(chunks)} if default4 and line["container_ok"] else {}
            line["pinned_chunks_checked"] = sorted(checked)
            line["pinned_chunks_ok"] = bool(checked) and all(v is True for v in checked.values())
            if not line["pinned_chunks_ok"] and default4:
                print(f"bench.py: config 4: pinned chunk streams differ from the real reference's: {checked}", file=sys.stderr)
                line["valid"] = False
            line["stage_ms"] = {k[3:]: round(v, 3) for k, v in ms.items()}
            line["schedule"] = {0: "sequential stages", 1: "pipelined (prefixes)", 2: "K3 in slices"}[int(st.pipelined)]
            line["one_stream_per_rank"] = {"value": round(synth.DICKENS_BYTES * world / float(sdt.item()) / 1e6, 3), "unit": "MB/s",
                                           "note": "every rank one dickens-sized stream (configs[1] shape), no exchange"}
            Yc = int(st.chain_symbols) or int(st.coded_symbols)
            kms = ms["ms_coder"]
            line["roofline"] = {"bound": "hbm", "kernel": "x3_ac2_kernel", "achieved": round(Yc * 16.5 / (kms * 1e-3) / 1e9, 3) if kms > 0 else None,
                                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(Yc * 16.5 / (kms * 1e-3) / HBM_PEAK, 7) if kms > 0 else None,
                                "traffic": None, "note": "rank 0's coder recurrence (16 chains side by side); see the N=1 line for the dominant-kernel analysis"}
            emit_line(line)
        dist.destroy_process_group()
        return

    # ---- N = 1: checks, roofline, secondary figures, CPU baseline ------------------------------------------------------------------
    stream = d_out[:out_len].cpu().numpy().tobytes()
    sha_ok = None
    if default_workload:
        man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest_sha.json")))["cfg2_full_english10192446_w64_t256"]
        sha_ok = hashlib.sha256(stream).hexdigest() == man["output_sha256"] and len(stream) == man["output_len"]
        assert sha_ok, "the GPU stream of the workload differs from the real reference's (tests/golden/manifest_sha.json)"

    S, H, Y, comp, N = int(st.steps), int(sum(list(st.events)[:3])), int(st.coded_symbols), out_len, args.bytes
    Yc = int(st.chain_symbols) or Y  # symbols the recurrence actually processes (no-op symbols are dropped)
    W = args.w * 1024
    # Algorithmic bytes per launch of each kernel family (what the algorithm must move):
    kernels = {
        "x3_ac2_kernel": {"ms": ms["ms_coder"], "alg_bytes": Yc * 16 + (Yc + 7) // 8 * 8},  # {cum, freq, magic, shift} in per symbol, one {lo, R} state out per 8 symbols
        "mode choice (fixed-point passes / x3_modes_kernel)": {"ms": ms["ms_modes"], "alg_bytes": H * (7 * 4 + 4)},  # 7 feature words in, mode out (per pass)
        "x3_parse_kernel": {"ms": ms["ms_parse"], "alg_bytes": 2 * N + 4 * S},        # bytes + m[] in, one token word out
        "scan (sort + lookup + x3_walk_kernel)": {"ms": ms["ms_scan"], "alg_bytes": S * W + N + comp},  # S*W + N + C
        "code features+emit (sorts/scans/CSB)": {"ms": ms["ms_features"] + ms["ms_emit"], "alg_bytes": None},
    }
    for k in kernels.values():
        k["GBps"] = round(k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9, 3) if k["alg_bytes"] and k["ms"] > 0 else None
        k["ms"] = round(k["ms"], 3)
    dom = max((k for k in kernels if kernels[k]["alg_bytes"]), key=lambda k: kernels[k]["ms"])
    # HBM bytes of the dominant kernel from SEPARATE rocprofv3 --pmc passes of this same command (tools/pmc_agg.py): only quoted when
    # that profile was taken with the same arguments and its kernel time agrees with this run's (else null: stale numbers are worse than none)
    traffic, traffic_src = None, None
    pmc_path = newest_profile("pmc_traffic.json")
    if pmc_path:
        pmc = json.load(open(pmc_path))
        same_args = pmc.get("bench_args") == {"bytes": args.bytes, "w": args.w, "t": args.t}
        e = pmc.get("kernels", {}).get(dom)
        if same_args and e and abs(e.get("kernel_ms_per_step", 0) - kernels[dom]["ms"]) <= 0.1 * kernels[dom]["ms"]:
            traffic = e["hbm_bytes_per_step"]
            traffic_src = f"profiles/{os.path.basename(pmc_path)}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (per step = all launches of the kernel; FETCH doubled per MI355X_MICROARCH.md)"
    path_bytes = S * W + N + comp  # the path's algorithmic bytes B_alg = S*W + N + C
    step_traffic = None  # HBM bytes of the WHOLE step (every kernel), same PMC profile, same staleness rule
    if pmc_path and traffic is not None:
        step_traffic = int(sum(k.get("hbm_bytes_per_step", 0) for k in pmc.get("kernels", {}).values()))
    line.update({
        "config": {"workload": f"dickens-like: {N} bytes of synthetic English-like text, ONE x3 stream, -w {args.w} -t {args.t}, bit-exact x3 code stream",
                   "window_kib": args.w, "max_match_count": args.t, "streams_per_gpu": 1},
        "ratio": round(N / comp, 4), "compressed_bytes": comp, "parse_steps": S, "coded_symbols": Y, "chain_symbols": Yc,
        "stream_sha256_equals_reference": sha_ok,
        "stage_ms": {k[3:]: round(v, 3) for k, v in ms.items()},
        "schedule": {0: "sequential stages",
                     1: "pipelined: parse / feature passes / coder recurrence overlap on three HIP streams, the coding stage re-run on growing prefixes; stage_ms are per-stage sums",
                     2: f"K3 in slices: the parse publishes checkpoints, every slice between two of them gets its features from carried model state ({int(st.coder_launches)} slices), "
                        "coder and bit emission continue per slice; parse / features / coder overlap on three HIP streams; stage_ms are per-stage sums "
                        "(features = records, ranks, context statistics; modes = mode chain, index / order-0 models, symbol assembly)"}[int(st.pipelined)],
        "mode_choice_iterations": int(st.mode_iters),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(kernels[dom]["alg_bytes"] / (kernels[dom]["ms"] * 1e-3) / HBM_PEAK, 7), "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes": kernels[dom]["alg_bytes"], "kernel_ms": kernels[dom]["ms"],
                     "launches_per_step": int(st.coder_launches) or 1,
                     # the compulsory-traffic floor N + C and the whole step's measured HBM traffic beside every fraction
                     "compulsory_bytes": N + comp, "step_traffic_bytes": step_traffic,
                     # SURVEY.md 8(d)'s single figure for the PATH, B_alg = S*W + N + C over the whole step, in the same object
                     "path_algorithmic_bytes": path_bytes, "path_achieved": round(path_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                     "frac_path": round(path_bytes / (ms_per_step * 1e-3) / HBM_PEAK, 5),
                     "frac_path_vs_copy_peak": round(path_bytes / (ms_per_step * 1e-3) / HBM_COPY_PEAK, 5),
                     "note": "`frac` / `achieved` price the OPERAND bytes of the dominant kernel (16 B of {cum, freq, magic, shift} per chain symbol + one 8-byte state per 8 symbols), not SURVEY 8(d)'s B_alg: that figure for the whole step is `frac_path` (vs 8.0 TB/s) / `frac_path_vs_copy_peak` (vs 6.29 TB/s) -- the sorted-n-gram scan never touches S*W bytes, so neither is a bandwidth statement.  Dominant kernel by time: ONE wavefront's dependent chain per stream, 13 scalar instructions per symbol at the 4-cycle single-wave issue rate, operands and chain states through the scalar cache; HBM is the stated bound, not the limiter (255 of 256 CUs idle: the x3 format fixes one adaptive coder chain per stream)"},
        "kernels": kernels,
        "path_roofline": {"algorithmic_bytes": path_bytes, "achieved_GBps": round(path_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                          "frac_of_hbm_peak": round(path_bytes / (ms_per_step * 1e-3) / HBM_PEAK, 5),
                          "note": "SURVEY.md 8(d): B_alg = S*W + N + C over the whole step (the sorted-n-gram scan never touches S*W bytes)"},
    })

    if not args.no_secondary:
        # host buffers: x3h_compress with the input in pageable host memory and the stream copied back (PCIe-inclusive; never `value`)
        ctx.compress(data, prm)
        hdt, hstream = timed(lambda: ctx.compress(data, prm), torch.cuda.synchronize)
        line["host_buffers"] = {"value": round(N / hdt / 1e6, 3), "unit": "MB/s", "ms": round(hdt * 1e3, 3), "copy_ms": round(ctx.last_stats.ms_copy, 3),
                                "note": "x3h_compress: H2D of the input + the step + D2H of the stream, wall clock"}

        # the SAME bytes cut into independent chunks (each its own x3 stream), one batch per chunk count.  The ratio pays for
        # every restart of the models; single stream = the `ratio` above.
        man_all = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest_sha.json")))
        sweep = []
        for nch in (16, 24, 32, 40, 48, 64, 80, 96, 128, 256, 512):
            cb = (N + nch - 1) // nch
            cdt, clens, cst, coff, d_cout, cstride = chunk_batch(ctx, d_in, N, cb, prm, dev, reps=4)  # (best of four: these are 7-25 ms calls)
            e = {"chunks": len(coff) - 1, "chunk_bytes": cb, "value": round(N / cdt / 1e6, 3), "unit": "MB/s", "ms": round(cdt * 1e3, 3),
                 "ratio": round(N / float(clens.sum()), 4),
                 "stage_ms": {"scan": round(cst.ms_scan, 3), "parse": round(cst.ms_parse, 3), "features": round(cst.ms_features, 3),
                              "modes": round(cst.ms_modes, 3), "coder": round(cst.ms_coder, 3), "emit": round(cst.ms_emit, 3)}}
            if nch in (40, 64, 128) and default_workload:  # first and last chunk of the TIMED call against the real reference's stream of that chunk alone
                hsl = lambda i: d_cout[i * cstride:i * cstride + int(clens[i])].cpu().numpy().tobytes()
                e["pinned_chunks_checked"], e["pinned_chunks_ok"] = pinned_pieces_ok(man_all, "english_like", hsl, names={f"csb_dickens_{nch}x_chunk{i}_w64_t256": i for i in (0, nch - 1)})
                if not e["pinned_chunks_ok"]:
                    e["valid"] = False
            if nch == 64:  # decoder (x3.c:285-353): the chunk streams decoded as one batch; and every 8th stream against the oracle-free check: round trip
                hout = d_cout.cpu().numpy()
                cstreams = [hout[i * cstride:i * cstride + int(clens[i])].tobytes() for i in range(len(coff) - 1)]
                caps = [int(coff[i + 1] - coff[i]) for i in range(len(coff) - 1)]
                back = ctx.decompress_chunks(cstreams, caps)
                dst = ctx.last_stats
                e["decode"] = {"kernel_ms": round(dst.ms_code + dst.ms_emit, 3), "chain_ms": round(dst.ms_code, 3), "bytes_stage_ms": round(dst.ms_emit, 3),
                               "value": round(N / ((dst.ms_code + dst.ms_emit) * 1e-3) / 1e6, 3), "unit": "MB/s",
                               "round_trip_ok": bool(b"".join(back) == data.tobytes())}
            sweep.append(e)
            del d_cout
        at1g = [e for e in sweep if e["value"] >= 1000.0 and e.get("valid", True)]
        line["chunked_same_bytes"] = {"single_stream_ratio": round(N / comp, 4), "sweep": sweep,
                                      "best": max(sweep, key=lambda e: e["value"])["value"],
                                      # the >= 1 GB/s point that keeps the most of the single stream's ratio (every restart of the models costs ratio)
                                      "best_ratio_at_1GBps": (lambda e: {"ratio": e["ratio"], "chunks": e["chunks"], "value": e["value"]})(max(at1g, key=lambda e: e["ratio"])) if at1g else None}

        if args.many_chunks_mib > 0:
            # aggregate figure: many independent streams in ONE batch, 256 KiB chunks of FRESH content (half English-like text, half Zipf(s=1)
            # bytes -- nothing repeated)
            mtot, mcb = args.many_chunks_mib << 20, 256 << 10
            q = mtot // 2
            mdata = synth.many_chunks_mix(mtot)
            d_min = torch.from_numpy(mdata).to(dev)
            mdt, mlens, mst, moff, d_mout, mstride = chunk_batch(ctx, d_min, mtot, mcb, prm, dev)
            line["many_chunks_batch"] = {
                "chunks": len(moff) - 1, "chunk_bytes": mcb, "total_bytes": mtot, "content": "fresh: 1/2 English-like text, 1/2 Zipf(s=1) bytes",
                "value": round(mtot / mdt / 1e6, 2), "unit": "MB/s", "ms": round(mdt * 1e3, 2), "ratio": round(mtot / float(mlens.sum()), 4),
                "stage_ms": {"scan": round(mst.ms_scan, 2), "parse": round(mst.ms_parse, 2), "features": round(mst.ms_features, 2),
                             "modes": round(mst.ms_modes, 2), "coder": round(mst.ms_coder, 2), "emit": round(mst.ms_emit, 2)}}
            if mtot == synth.MANY_CHUNKS_MIB << 20 and (args.w, args.t) == (64, 256):
                # bit-exactness of the TIMED form: fourteen chunks -- first / middle / last of the text half and of the Zipf half, both sides of the
                # text -> Zipf boundary -- against the real reference's `x3 -z -w 64 -t 256` of that chunk alone (x3.c:372-434,593-611)
                mcheck, mok = pinned_pieces_ok(man_all, "many_chunks_mix", lambda c: d_mout[c * mstride:c * mstride + int(mlens[c])].cpu().numpy().tobytes(), mcb)
                line["many_chunks_batch"].update({"pinned_chunks_checked": mcheck, "pinned_chunks_ok": mok})
                if not mok:
                    line["many_chunks_batch"]["valid"] = False
            # K1 of this batch is the one HBM-bound hand-written kernel family of the path (scan3.hip: per-chunk radix sort + level tests, dense classes
            # refined by the same workgroup): algorithmic bytes = 80 per list element (8 keys out + 4 passes x 16 + 8 for the level-4 re-read), n + 3 elements per chunk
            k1_alg = 80 * (mtot + 3 * (len(moff) - 1))
            line["many_chunks_batch"]["k1_roofline"] = {"bound": "hbm", "kernel": "K1 stage = x3_segscan_kernel + x3_segrefine_kernel + x3_walk_kernel", "algorithmic_bytes": k1_alg,
                                                        "divided_by": "stage_ms.scan (HIP events around the whole K1 stage on the library's stream)", "stage_ms": round(mst.ms_scan, 2),
                                                        "achieved": round(k1_alg / (mst.ms_scan * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                                        "frac": round(k1_alg / (mst.ms_scan * 1e-3) / HBM_PEAK, 4),
                                                        "note": "x3_segscan_kernel alone (kernel trace, profiles/rNN_many_chunks_mix_kernel_stats.csv): frac_kernel below; its PMC traffic: hbm_traffic.by_family_GB"}
            kpath = newest_profile("many_chunks_mix_kernel_stats.csv")
            if kpath:  # the kernel's own average duration from the committed kernel trace of the same batch
                import csv
                for row in csv.DictReader(open(kpath)):
                    if row.get("Name", "").startswith("x3_segscan_kernel"):
                        kms = float(row["AverageNs"]) * 1e-6
                        line["many_chunks_batch"]["k1_roofline"].update({"kernel_ms": round(kms, 3), "frac_kernel": round(k1_alg / (kms * 1e-3) / HBM_PEAK, 4),
                                                                         "kernel_ms_source": f"profiles/{os.path.basename(kpath)}"})
            tdt, tlens, tst, _, d_tout, _ = chunk_batch(ctx, d_min[:q], q, mcb, prm, dev)  # the text half alone (round 1 measured tiled text)
            line["many_chunks_batch"]["text_only"] = {"total_bytes": q, "value": round(q / tdt / 1e6, 2), "unit": "MB/s", "ratio": round(q / float(tlens.sum()), 4)}
            # decoder (x3.c:285-353), one wavefront per stream: the whole fresh batch decoded back as ONE batch, streams and bytes resident in HBM
            def decode_leg(d_streams, stride, lens, cb):
                nst = len(lens)
                d_cmp = torch.cat([d_streams[i * stride:i * stride + int(lens[i])] for i in range(nst)])   # back to back (whole 32-bit words each)
                ioff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
                ooff = np.arange(0, (nst + 1) * cb, cb, dtype=np.uint64)
                d_back = torch.empty(nst * cb, dtype=torch.uint8, device=dev)
                best = None
                for _ in range(2):
                    ddt, (dl, dst) = timed(lambda: ctx.decompress_chunks_dev(d_cmp.data_ptr(), ioff, d_back.data_ptr(), ooff), torch.cuda.synchronize)
                    if best is None or ddt < best[0]:
                        best = (ddt, dst.ms_code + dst.ms_emit, dst.ms_code)
                ok = bool(int(dl.sum()) == mtot and torch.equal(d_back[:mtot], d_min))
                return {"streams": nst, "ms": round(best[0] * 1e3, 2), "kernel_ms": round(best[1], 2), "chain_ms": round(best[2], 2), "value": round(mtot / best[0] / 1e6, 2), "unit": "MB/s", "round_trip_ok": ok}
            line["many_chunks_batch"]["decode"] = decode_leg(d_mout, mstride, mlens, mcb)
            line["many_chunks_batch"]["decode"]["note"] = "x3h_decompress_chunks_dev: stage 1 = one chain per stream writes a tag per parse step (chain_ms), stage 2 = tags -> bytes over the whole chip; the batch rate is streams in flight x the per-stream rate: the same bytes as 64 KiB streams below"
            fdt, flens, fst, foff, d_fout, fstride = chunk_batch(ctx, d_min, mtot, 64 << 10, prm, dev, reps=1)
            line["many_chunks_batch"]["decode_64KiB_streams"] = dict(decode_leg(d_fout, fstride, flens, 64 << 10), ratio=round(mtot / float(flens.sum()), 4),
                                                                 compress_value=round(mtot / fdt / 1e6, 2))
            del d_fout
            tpath = newest_profile("many_chunks_pmc_traffic.json")
            if tpath:
                t = json.load(open(tpath))
                if t.get("total_bytes") == mtot and abs(t.get("batch_ms", 0) - mdt * 1e3) <= 0.15 * mdt * 1e3:
                    line["many_chunks_batch"]["hbm_traffic"] = {"GB_per_batch": t["total"]["GB"], "bytes_per_input_byte": t["total"]["bytes_per_input_byte"],
                                                                "avg_TBps_over_kernel_time": t["total"]["avg_TBps"], "frac_of_hbm_peak": round(t["total"]["avg_TBps"] * 1e12 / HBM_PEAK, 3),
                                                                "source": f"profiles/{os.path.basename(tpath)} (separate rocprofv3 --pmc passes of tools/many_chunks_check.py on the same batch)",
                                                                "by_family_GB": {k: v["hbm_GB"] for k, v in list(t.get("groups", {}).items())[:8]}}
            del d_min, d_mout, d_tout
            # data with DENSE classes (sparse 16-bit samples: thousands of repeats of a gram inside every window): K1 refines such classes byte
            # by byte instead of sweeping them (scan2.hip) -- bounded, but the scan dominates
            ddata = synth.dense_batch()
            d_din = torch.from_numpy(ddata).to(dev)
            ddt, dlens, dst_, _, d_dout, dstride = chunk_batch(ctx, d_din, ddata.size, mcb, prm, dev, reps=1)
            line["many_chunks_dense_classes"] = {"content": "mr-like 16-bit samples (zero background, sparse noise)", "total_bytes": int(ddata.size), "chunks": ddata.size // mcb,
                                                 "value": round(ddata.size / ddt / 1e6, 2), "unit": "MB/s", "ratio": round(ddata.size / float(dlens.sum()), 4),
                                                 "stage_ms": {"scan": round(dst_.ms_scan, 2), "parse": round(dst_.ms_parse, 2), "code": round(dst_.ms_code, 2)}}
            if (args.w, args.t) == (64, 256):
                dcheck, dok = pinned_pieces_ok(man_all, "dense_batch", lambda c: d_dout[c * dstride:c * dstride + int(dlens[c])].cpu().numpy().tobytes(), mcb)
                line["many_chunks_dense_classes"].update({"pinned_chunks_checked": dcheck, "pinned_chunks_ok": dok})
                if not dok:
                    line["many_chunks_dense_classes"]["valid"] = False
            del d_din, d_dout

        # one GPU's share of config 4 (16 chunks x 8 MiB of the Zipf stream): what every rank does at N > 1, without the gather
        zdata = synth.zipf_bytes(16 * CHUNK4)
        d_zin = torch.from_numpy(zdata).to(dev)
        zdt, zlens, zst, _, d_zout, zstride = chunk_batch(ctx, d_zin, 16 * CHUNK4, CHUNK4, prm, dev, reps=1)
        # the 16-chunk batch takes its own schedule (pipelined, segment-wise bit emission): its pinned chunks are compared with the real
        # reference's streams HERE, on the output of the timed call (bit-exactness asserted in the same run)
        z_ok = {c: pinned_chunk_ok(man_all, c, d_zout[c * zstride:c * zstride + int(zlens[c])].cpu().numpy().tobytes()) for c in PINNED4 if c < 16} \
            if (args.w, args.t) == (64, 256) else {}
        z_bad = sorted(c for c, v in z_ok.items() if v is not True)
        if z_bad:
            print(f"bench.py: config 4 share: chunk stream(s) {z_bad} differ from the real reference's -- leg marked invalid", file=sys.stderr)
        line["config4_share_per_gpu"] = {"valid": False, "chunks_that_differ_from_the_reference": z_bad} if z_bad else {"chunks": 16, "chunk_bytes": CHUNK4, "value": round(16 * CHUNK4 / zdt / 1e6, 3), "unit": "MB/s", "ms": round(zdt * 1e3, 2),
                                         "ratio": round(16 * CHUNK4 / float(zlens.sum()), 4), "pipelined": int(zst.pipelined),
                                         "pinned_chunks_checked": sorted(z_ok), "pinned_chunks_ok": bool(z_ok) and not z_bad,
                                         "stage_ms": {"scan": round(zst.ms_scan, 2), "parse": round(zst.ms_parse, 2), "features": round(zst.ms_features, 2),
                                                      "modes": round(zst.ms_modes, 2), "coder": round(zst.ms_coder, 2)}}
        del d_zin, d_zout

        def sha_matches(name, stream):
            """True / False against the REAL reference's sha256 of this case (tests/golden/manifest_sha.json), None when the case is not pinned"""
            e = man_all.get(name)
            return None if e is None else bool(len(stream) == e["output_len"] and hashlib.sha256(stream).hexdigest() == e["output_sha256"])

        # config 3: twelve independent streams with the Silesia sizes (every third Zipf bytes, the others English-like text), -w 256 -t 1024,
        # ONE batch; every stream's sha256 is compared with the real reference's inside the run
        if not args.no_config35:
            try:
                prm3 = _lib.make_params(w_kib=256, t=1024)
                names3 = list(synth.SILESIA)
                parts = [synth.config3_part(i) for i in range(len(names3))]
                sizes3 = [int(p.size) for p in parts]
                tot3 = sum(sizes3)
                d_3in = torch.from_numpy(np.concatenate(parts)).to(dev)
                off3 = np.concatenate([[0], np.cumsum(sizes3)]).astype(np.uint64)
                stride3 = (max(sizes3) + (max(sizes3) >> 2) + 4096 + 3) & ~3
                d_3out = torch.empty(stride3 * len(sizes3), dtype=torch.uint8, device=dev)
                best3 = None
                for it in range(2):
                    dt3, (lens3, st3) = timed(lambda: ctx.compress_chunks_dev(d_3in.data_ptr(), off3, prm3, d_3out.data_ptr(), stride3), torch.cuda.synchronize)
                    if best3 is None or dt3 < best3[0]:
                        best3 = (dt3, st3)
                h3 = d_3out.cpu().numpy()
                oks = [sha_matches(f"cfg3_full_{i:02d}_{nm}{sizes3[i]}_w256_t1024", h3[i * stride3:i * stride3 + int(lens3[i])].tobytes()) for i, nm in enumerate(names3)]
                bad3 = [names3[i] for i, o in enumerate(oks) if o is False]
                if bad3:  # a stream that differs from the reference's voids THIS leg (its number is not reported); the headline has its own hard check above
                    print(f"bench.py: config 3: stream(s) {bad3} differ from the real reference's -- leg marked invalid", file=sys.stderr)
                line["config3_full"] = {"valid": False, "streams_that_differ_from_the_reference": bad3} if bad3 else {"streams": len(sizes3), "total_bytes": tot3, "args": "-w 256 -t 1024", "value": round(tot3 / best3[0] / 1e6, 3), "unit": "MB/s",
                                        "ms": round(best3[0] * 1e3, 1), "ratio": round(tot3 / float(lens3.sum()), 4), "pipelined": int(best3[1].pipelined),
                                        "parse_steps": int(best3[1].steps), "streams_sha256_equal_reference": sum(1 for o in oks if o), "streams_not_pinned": sum(1 for o in oks if o is None),
                                        "stage_ms": {"scan": round(best3[1].ms_scan, 1), "parse": round(best3[1].ms_parse, 1), "features": round(best3[1].ms_features, 1),
                                                     "modes": round(best3[1].ms_modes, 1), "coder": round(best3[1].ms_coder, 1)},
                                        "content": "synthetic stand-ins with the 12 Silesia file sizes (synth.config3_part)"}
                del d_3in, d_3out, h3, parts

                # config 5: mr-sized stream of mr-like 16-bit samples, -w 512 -t 4096, compress + GPU decode round trip
                prm5 = _lib.make_params(w_kib=512, t=4096)
                d5 = synth.mr_like(synth.MR_BYTES)
                d_5in = torch.from_numpy(d5).to(dev)
                stride5 = (2 * d5.size + 4096 + 3) & ~3
                d_5out = torch.empty(stride5, dtype=torch.uint8, device=dev)
                off5 = np.array([0, d5.size], dtype=np.uint64)
                ctx.compress_chunks_dev(d_5in.data_ptr(), off5, prm5, d_5out.data_ptr(), stride5)
                dt5, (lens5, st5) = timed(lambda: ctx.compress_chunks_dev(d_5in.data_ptr(), off5, prm5, d_5out.data_ptr(), stride5), torch.cuda.synchronize)
                s5 = d_5out[:int(lens5[0])].cpu().numpy().tobytes()
                ok5 = sha_matches(f"cfg5_full_mr{d5.size}_w512_t4096", s5)
                if ok5 is False:
                    print("bench.py: config 5: the GPU stream differs from the real reference's -- leg marked invalid", file=sys.stderr)
                d_5back = torch.empty(d5.size, dtype=torch.uint8, device=dev)
                ioff5 = np.array([0, (len(s5) + 3) & ~3], dtype=np.uint64)
                ddt5, (dl5, dst5) = timed(lambda: ctx.decompress_chunks_dev(d_5out.data_ptr(), ioff5, d_5back.data_ptr(), off5), torch.cuda.synchronize)
                line["config5_round_trip"] = {"bytes": int(d5.size), "args": "-w 512 -t 4096", "compress_value": round(d5.size / dt5 / 1e6, 3), "compress_ms": round(dt5 * 1e3, 1),
                                              "decode_value": round(d5.size / ddt5 / 1e6, 3), "decode_ms": round(ddt5 * 1e3, 1), "decode_chain_ms": round(dst5.ms_code, 1), "decode_bytes_stage_ms": round(dst5.ms_emit, 2),
                                              "decode_ns_per_step": round(dst5.ms_code * 1e6 / max(int(st5.steps), 1), 1), "unit": "MB/s", "ratio": round(d5.size / len(s5), 4),
                                              "parse_steps": int(st5.steps), "stream_sha256_equals_reference": ok5,
                                              "round_trip_ok": bool(int(dl5[0]) == d5.size and torch.equal(d_5back, d_5in)),
                                              "content": "mr-like 16-bit samples (synth.mr_like), the size of Silesia 'mr'"}
                # the reference's decoder beside the GPU's (x3.c:641-645 prints its own elapsed time), on a BOUNDED sample: the first MiB of this input, coded on the
                # GPU with config 5's parameters (the stream a reference encoder would write), decoded by oracle/_ref/x3 -d on one host core and by the GPU
                if not args.no_cpu:
                    samp = d5[:1 << 20]
                    s_samp = ctx.compress(samp, prm5)
                    rd = run_reference_decode(s_samp)
                    ctx.decompress(s_samp, samp.size)
                    gdt, gback = timed(lambda: ctx.decompress(s_samp, samp.size), torch.cuda.synchronize)
                    if rd is not None:
                        line["config5_round_trip"]["cpu_baseline_decode"] = {
                            "value": round(samp.size / rd[0] / 1e6, 5), "unit": "MB/s", "cores": 1, "kind": "reference", "seconds": round(rd[0], 3),
                            "sample": f"first {samp.size} bytes of config 5's input as one -w 512 -t 4096 stream ({len(s_samp)} bytes), `x3 -d`'s own 'elapsed time' (x3.c:641-645)",
                            "reference_output_equals_input": bool(rd[1] == samp.tobytes()),
                            "gpu_decode_same_stream": {"value": round(samp.size / gdt / 1e6, 3), "unit": "MB/s", "ms": round(gdt * 1e3, 2), "round_trip_ok": bool(gback == samp.tobytes()),
                                                       "note": "x3h_decompress, host buffers (H2D + D2H included)"}}
                if ok5 is False or not line["config5_round_trip"]["round_trip_ok"]:
                    line["config5_round_trip"] = {"valid": False, "stream_sha256_equals_reference": ok5, "round_trip_ok": line["config5_round_trip"]["round_trip_ok"]}
                del d_5in, d_5out, d_5back
            except Exception as e:  # a secondary leg must not take the line with it
                print(f"bench.py: config 3 / 5 legs failed: {e!r}", file=sys.stderr)
                line.setdefault("config3_full", {"valid": False, "error": repr(e)})
                line.setdefault("config5_round_trip", {"valid": False, "error": repr(e)})

    if not args.no_cpu:
        # CPU baseline, this box's host cores: (1) one core on a bounded prefix of THE workload (the reference is single-threaded);
        # (2) all cores, one reference process per core on distinct 128 KiB pieces (what the chunked configs can use: chunks are independent)
        sec, ref_out, kind = run_reference(data[:args.cpu_sample].tobytes(), args.w, args.t)
        gpu_sample = ctx.compress(data[:args.cpu_sample], prm)
        line["cpu_baseline"] = {"value": round(args.cpu_sample / sec / 1e6, 5), "unit": "MB/s", "cores": 1, "kind": kind,
                                "sample": f"first {args.cpu_sample} bytes of the workload, -w {args.w} -t {args.t}, x3's own 'elapsed time' (x3.c:597-601)",
                                "seconds": round(sec, 3), "bit_exact_vs_gpu_on_sample": bool(gpu_sample == ref_out)}
        if not args.no_secondary:
            ncpu = os.cpu_count() or 1
            piece = 128 * 1024
            pieces = [data[i * piece * 3:i * piece * 3 + piece].tobytes() for i in range(ncpu)]
            t1 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=ncpu) as ex:
                res = list(ex.map(lambda s: run_reference(s, args.w, args.t), pieces))
            wall = time.perf_counter() - t1
            line["cpu_baseline_all_cores"] = {"value": round(ncpu * piece / wall / 1e6, 5), "unit": "MB/s", "cores": ncpu, "nproc": ncpu, "kind": res[0][2],
                                              "sample": f"{ncpu} reference processes side by side, each on its own {piece}-byte piece of the workload (independent chunks), wall clock",
                                              "seconds": round(wall, 3)}
    emit_line(line)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
