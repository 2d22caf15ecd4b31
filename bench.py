#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

One step = one full render of the workload: clear the linear accumulation buffer, trace every sample of every
pixel (each rank its share of the sample indices), reduce the ranks' float4 accumulation buffers to rank 0
over RCCL (N > 1 only), and run the gamma/present resolve.  Scene, camera and buffers are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X FP32 vector peak (256 CUs x 256 flop/clk x 2.4 GHz)
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense FP16 matrix peak
FLOPS_PER_TEST = 23            # one ray-sphere test as the reference writes it, up to the discriminant early-out
MFMA_FLOPS_PER_BLOCK = 8 * 2 * 32 * 32 * 16  # one 64-sphere block x 64 rays: 8 v_mfma_f32_32x32x16_f16


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--scene", default="cover")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--spp", type=int, default=1000)
    p.add_argument("--depth", type=int, default=50)
    p.add_argument("--chunk", type=int, default=-1, help="samples per work item (-1 = library default)")
    p.add_argument("--tail-chunk", type=int, default=-1, help="samples per work item at the end of a launch (-1 = library default)")
    p.add_argument("--tail-samples", type=int, default=-1, help="samples per pixel handed out in such short items (-1 = library default)")
    p.add_argument("--fetch", type=int, default=0, help="0 automatic, 1 scalar cache, 2 per-wave streamed tiles, 3 lifted tiles per workgroup (matrix cores)")
    p.add_argument("--filter", type=int, default=0, help="0 automatic (matrix cores where the scene allows), 1 general vector "
                   "filter, 2 vector filters only, 3 matrix cores whatever the scene size")
    p.add_argument("--blocks-per-cu", type=int, default=0)
    p.add_argument("--partition", default="sample", choices=["sample", "tile"],
                   help="N > 1: split the sample indices over the ranks (default), or rows of 64-pixel tiles (comparison)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo is for rehearsing N > 1 on fewer GPUs")
    p.add_argument("--cpu-spp", type=int, default=64, help="spp of the sub-sampled CPU baseline run")
    p.add_argument("--cpu-stride", type=int, default=8, help="pixel stride of the sub-sampled CPU baseline run")
    p.add_argument("--cpu-runs", type=int, default=3, help="repetitions of the CPU baseline run (the best is reported)")
    p.add_argument("--valu-steps", type=int, default=2, help="N = 1: extra timed steps with the vector sphere filter "
                   "(the design point the FP32 vector peak applies to); 0 = skip")
    p.add_argument("--dump", default=None, help="rank 0 writes the last step's reduced linear sums and resolved image "
                   "to this .npz file")
    return p.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(rt, ge, args):
    """The oracle (a plain-C port of the reference's kernel) on the host cores, on a stratified sub-sample of the
    same workload: every `stride`-th pixel in x and y at `cpu_spp` samples per pixel, `cpu_runs` times."""
    ora = ge.load_oracle()
    ora.rto_dispatch.restype = None
    ora.rto_dispatch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    import numpy as np
    objs, n = rt.scene(args.scene)
    u = rt.config_uniforms(args.scene, args.width, args.height, args.cpu_spp, args.depth, n)
    npx = len(range(0, args.width, args.cpu_stride)) * len(range(0, args.height, args.cpu_stride))
    samples = npx * args.cpu_spp
    times = []
    for _ in range(max(1, args.cpu_runs)):
        tex = np.zeros((args.height, args.width, 4), np.float32)
        t0 = time.perf_counter()
        ora.rto_dispatch(tex.ctypes.data, args.width, args.height, C.byref(u), objs, 0, args.cpu_stride, None)
        times.append(time.perf_counter() - t0)
    dt = min(times)
    return {
        "value": samples / dt * 1e-6,
        "unit": "Msamples*px/s",
        "cores": int(ora.rto_num_threads()),
        "kind": "port",
        "cpu": cpu_model(),
        "runs_s": [round(t, 3) for t in times],
        "sample": "every %dth pixel in x and y of the %dx%d %s frame at %d spp, depth %d (%d samples per run, best of %d runs)"
                  % (args.cpu_stride, args.width, args.height, args.scene, args.cpu_spp, args.depth, samples, len(times)),
    }


def main():
    args = parse()
    import torch  # before the HIP library, so both share one HIP runtime
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d: launch N > 1 with python -m torch.distributed.run "
                         "--nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (local_rank, ndev))
    local_rank = local_rank % ndev  # only reachable with --backend gloo: ranks share GPUs for a rehearsal
    torch.cuda.set_device(local_rank)
    # RT_BENCH_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): a one-rank job takes the multi-rank code
    # path too -- process group, reduce, barrier -- so that the RCCL calls can be exercised on a box with one GPU
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    rt = ge.load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_partition", os.path.join(ROOT, "ray-tracing_amd", "partition.py"))
    part = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(part)

    W, H, spp = args.width, args.height, args.spp
    # One stream of our own for everything: the library's launches, torch's clears and copies, and the collective
    # (torch orders a collective against the stream that is current when it is called).  The default stream has the
    # null handle, which the library reads as "create a private stream" -- nothing torch does would be ordered with it.
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)
    r = rt.MetalRenderer(device=local_rank, stream=stream.cuda_stream)
    r.setupTexture(W, H)
    if args.chunk >= 0:
        r.set_option(rt.RT_OPT_CHUNK, args.chunk)
    if args.tail_chunk >= 0:
        r.set_option(rt.RT_OPT_TAIL_CHUNK, args.tail_chunk)
    if args.tail_samples >= 0:
        r.set_option(rt.RT_OPT_TAIL_SAMPLES, args.tail_samples)
    r.set_option(rt.RT_OPT_SCENE_FETCH, args.fetch)
    r.set_option(rt.RT_OPT_FILTER, args.filter)
    r.set_option(rt.RT_OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    objs, n = rt.scene(args.scene)
    u = rt.config_uniforms(args.scene, W, H, spp, args.depth, n)
    r.setScene(objs, n)  # resident before the timed region
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    if args.partition == "sample":
        begin, end = part.ray_range(rank, world, spp)
    else:
        # every rank traces all samples of its own rows; the other rows of its buffer stay zero, so the same SUM
        # reduce acts as the gather of disjoint tiles
        begin, end = 0, spp
        y0, y1 = part.tile_rows(rank, world, H)
        if y1 <= y0:
            begin = end = 0  # more ranks than tile rows: nothing to trace, the buffer stays zero
        else:
            u.tileY, u.tileHeight = y0, y1 - y0

    def step():
        r.accumClear(accum.data_ptr())
        if end > begin:
            r.render(u, None, 0, begin, end, accum.data_ptr())
        if use_dist:
            if args.backend == "nccl":
                dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI; `stream` is current, so it follows the trace
            else:
                host = accum.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                accum.copy_(host)
        if rank == 0:
            r.resolve(spp, accum.data_ptr())

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        # build the communicator before anything is timed, whatever --warmup says
        dist.all_reduce(torch.zeros(1, device="cuda" if args.backend == "nccl" else "cpu"))
    for _ in range(args.warmup):
        step()
    fence()
    r.traceTimes()  # drop the warm-up launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    trace_ms = r.traceTimes()
    st = r.lastStats()
    if args.dump and rank == 0:
        import numpy as np
        torch.cuda.synchronize()
        np.savez(args.dump, accum=accum.cpu().numpy(), image=r.readTexture(), begin=begin, end=end, chunk=st.chunk,
                 chunk_count=st.chunk_count, tail_chunk=st.tail_chunk)

    # N = 1: the same workload with the vector sphere filter (no matrix instructions), a few steps: the design point
    # whose arithmetic the FP32 vector peak prices
    valu = None
    if world == 1 and args.valu_steps > 0 and st.scene_path in (3, 4, 5):
        r.set_option(rt.RT_OPT_FILTER, 2)
        step()
        fence()
        r.traceTimes()
        t1 = time.perf_counter()
        for _ in range(args.valu_steps):
            step()
        fence()
        dt2 = time.perf_counter() - t1
        ms2 = r.traceTimes()
        st2 = r.lastStats()
        avg2 = sum(ms2) / max(1, len(ms2))
        tf2 = FLOPS_PER_TEST * st2.tests / (avg2 * 1e-3) * 1e-12 if avg2 > 0 else 0.0
        valu = {
            "what": "same workload, RT_OPT_FILTER=2: sphere filter on the vector ALU (9 FMA-class instructions per test), "
                    "no matrix instructions; %d timed steps" % args.valu_steps,
            "value": W * H * spp / (dt2 / args.valu_steps) * 1e-6, "unit": "Msamples*px/s",
            "ms_per_step": dt2 / args.valu_steps * 1e3, "avg_launch_ms": avg2,
            "achieved": tf2, "peak": FP32_VALU_PEAK_TFLOPS, "unit_achieved": "TFLOP/s (23 flops per ray-object test)",
            "frac": tf2 / FP32_VALU_PEAK_TFLOPS, "sphere_filter_path": st2.scene_path,
        }
        r.set_option(rt.RT_OPT_FILTER, args.filter)

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        total_samples = W * H * spp
        avg_trace_ms = sum(trace_ms) / max(1, len(trace_ms))
        flops_per_launch = FLOPS_PER_TEST * st.tests
        algorithmic = flops_per_launch / (avg_trace_ms * 1e-3) * 1e-12 if avg_trace_ms > 0 else 0.0
        matrix = st.scene_path in (3, 4, 5)
        # Executed matrix work of one launch, counted by the kernel itself: 64-sphere blocks x 64 rays that the waves
        # put through the filter (lanes without a ray ride along), 8 v_mfma_f32_32x32x16_f16 each.
        mfma_flops = MFMA_FLOPS_PER_BLOCK * st.matrix_blocks
        mfma_tflops = mfma_flops / (avg_trace_ms * 1e-3) * 1e-12 if matrix and avg_trace_ms > 0 else 0.0
        # HBM traffic of the trace kernel is not measurable from inside this process: it is replayed from the counter
        # run committed under profiles/, and only while that run was taken on this workload and this kernel source.
        traffic, traffic_src = None, None
        import glob
        runs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_trace_hbm_bytes.json")))   # the latest round's counter run
        pmc_path = runs[-1] if runs else ""
        if world == 1 and pmc_path:
            try:
                rec = json.load(open(pmc_path))
                if (rec.get("workload") == "%s %dx%d %dspp" % (args.scene, W, H, spp)
                        and rec.get("kernel_source_id") == ge.kernel_source_id()):
                    traffic = rec.get("bytes_per_launch")
                    traffic_src = "replayed from profiles/%s (rocprofv3 --pmc run of this workload, kernel source %s)" % (os.path.basename(pmc_path), rec.get("kernel_source_id"))
            except Exception:
                traffic = None
        if matrix:
            roofline = {
                "bound": "mfma",
                "achieved": mfma_tflops,
                "peak": F16_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": mfma_tflops / F16_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "trace_kernel (rank 0)",
                "note": "achieved = executed FP16 matrix flops of the conservative sphere filter (kernel-counted 64x64 "
                        "blocks x 8 instructions x 32768 flop, a block at half width counting as half) / launch time, "
                        "against the dense FP16 matrix peak; the kernel is bound by SIMD issue shared between matrix and "
                        "vector instructions, so the rest of the time is vector work (sign extraction, exact hit tests, "
                        "shading).  Rays that cannot touch the box of the spheres the filter stands for are not put "
                        "through it (RT_OPT_NARROW): executed matrix work per closest-hit search is matrix_blocks_per_search "
                        "of the scene's blocks_per_search_at_full_width, so frac falls where value rises",
                "mfma_flops_per_launch": mfma_flops,
                "matrix_blocks_per_search": st.matrix_blocks / max(1, st.wave_trips),
                "blocks_per_search_at_full_width": (st.tests // max(1, st.segments) + 63) // 64 if st.scene_path == 3 else None,
            }
        else:
            roofline = {
                "bound": "valu_fp32",
                "achieved": algorithmic,
                "peak": FP32_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": algorithmic / FP32_VALU_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "trace_kernel (rank 0)",
                "note": "achieved counts the reference's 23 flops per ray-object test; the sphere filter runs on the vector ALU",
            }
        roofline.update({
            "sphere_filter": {0: "vector, records staged in LDS", 1: "vector, scalar-cache fetch",
                              2: "vector, streamed tiles", 3: "matrix cores (v_mfma_f32_32x32x16_f16)",
                              4: "matrix cores, lifted tiles streamed through LDS",
                              5: "matrix cores, lifted tiles read in place by the waves whose rays can touch them"}[st.scene_path],
            # the reference's brute-force loop priced at 23 flops per ray-object test: a rate of useful work, not a
            # fraction of any pipe once the filter runs on the matrix cores
            "algorithmic_tflops": algorithmic,
            "algorithmic_flops_per_launch": flops_per_launch,
            "avg_launch_ms": avg_trace_ms,
            "segments_per_sample": st.segments / max(1, st.samples),
            "wave_trips": st.wave_trips,
            "valu_design_point": valu,
        })
        out = {
            "metric": "Msamples*px/s (whole job), RTiOW cover 1080p; roofline: fraction of the dense FP16 matrix peak (default kernel) and of the FP32 VALU peak (vector design point)",
            "value": total_samples / (ms_per_step * 1e-3) * 1e-6,
            "unit": "Msamples*px/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (generated %s scene, fixed seed; random-free camera)" % args.scene,
            "config": {
                "workload": "%s scene (%d spheres) %dx%d, %d spp, depth %d, one pass" % (args.scene, n, W, H, spp, args.depth),
                "arithmetic": "hit decisions and radiance bit-identical to the uncontracted IEEE oracle",
                "samples_per_rank": end - begin,
                "parallelism": ("sample-index split x%d, RCCL reduce of float4 accumulation buffers" % world)
                               if args.partition == "sample" else
                               ("rows of 64-pixel tiles split x%d, RCCL reduce as gather" % world),
                "kernel_source_id": ge.kernel_source_id(),
            },
            "roofline": roofline,
        }
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rt, ge, args)
        print(json.dumps(out), flush=True)
    r.close()


if __name__ == "__main__":
    main()
