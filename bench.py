"""Headline benchmark: Msamples/s of the path-integrator hot path on the Cornell box,
1920x1080 @ 512 spp, diffuse BSDFs (BASELINE.json configs[1]), N GPUs of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--width --height --spp]

A "step" is one complete render of the workload (scene + BVH already resident in
HBM; the timed region is mi_render + the film reduce). N > 1 = one rank per GPU: either launched under
torch.distributed.run (RANK / WORLD_SIZE in the environment), or `python bench.py --gpus N` alone, which re-executes
itself under torch.distributed.run with N ranks (spawn_ranks). Pixel tiles (spiral blocks) are sharded
round-robin over ranks (the N-GPU film equals the 1-GPU film), each rank
splats into a private full-size float32 film and one RCCL reduce(sum) to rank 0 closes
the step. `--shard passes` (never chosen automatically) renders the reference's
samples_per_pass = spp / N job instead, one pass per rank; the JSON line names the
partition in `shard` and `config.parallelism`.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`
(dominant kernel, HIP-event timed inside the library on its own stream) and
`cpu_baseline` (the scalar_rgb oracle on the host cores, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
# algorithmic bytes per path segment per kernel (sum = 280 B)
B_SHADE = 196.0              # state R+W 96, hit R 16, ext ray W 28, shadow ray+contribution W 44, contribution R 12
B_TRACE_CLOSEST = 44.0       # ray R 28, hit W 16
B_TRACE_ANY = 40.0           # shadow ray R 32, visibility W+R 8
B_SPLAT = 320.0              # per finished sample: 4x4 texels x 5 channels x 4 B


def kernel_src_sha16():
    """sha256 (first 16 hex digits) over the kernel sources (everything under mitsuba2_amd/csrc, sorted by path: leaf arithmetic,
    device kernels, LBVH builder AND miwave.hip, which picks the kernel variant, the waves per SIMD, the grids and the shade
    vote): what a committed PMC profile must have been taken on for its numbers to be quoted in the JSON line"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "mitsuba2_amd", "csrc")
    for d, _, files in sorted(os.walk(base)):
        for f in sorted(files):
            if f.endswith((".h", ".hip")):
                h.update(os.path.relpath(os.path.join(d, f), base).encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def effective_cpus():
    """CPUs this process can actually run on: the affinity mask, capped by the cgroup CPU quota (v2 cpu.max, v1 cfs quota).
    os.cpu_count() reports the machine (256 hardware threads on the GPU box) whatever the lease allows — round 2 quoted it as
    `cores` while 256 oracle threads ran 6x one thread: the lease is a handful of CPUs."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except Exception:
            quota = None
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(O, np, scene, make_integrator, sensor, W, H, SPP):
    """The scalar_rgb oracle (kind "port": the reference binary cannot be built) on the host CPUs this
    process may use, on a bounded sample of the same job: whole spiral blocks (32x32 pixels, all spp), centre-most first.
    Three timings: one thread (>= 1 M samples), `cores` threads (the quoted value), and 2 x `cores` threads — if the last one
    is faster than the second by more than 15 % the lease has more CPUs than the mask / quota admit and `cores` is raised to
    what was used (the figure must name the threads that produced it)."""
    cores = effective_cpus()
    one = make_integrator().render_job(sensor)
    per_block = 1024 * SPP

    def run(threads, nblocks):
        _, _, st = O.render(scene.desc(), one, threads=threads, want_f64=False, only_blocks=np.arange(nblocks, dtype=np.uint32))
        return st.samples / st.seconds / 1e6, st

    n1 = max(2, -(-1_000_000 // per_block))                  # >= 1 M samples for the one-thread figure
    v1, st1 = run(1, n1)
    # ~10 s of work for `cores` threads at the one-thread rate, whole blocks, at least 3 per thread
    nb = int(max(3 * cores, min(2040, 10.0 * v1 * 1e6 * cores / per_block)))
    vc, stc = run(cores, nb)
    used, best, stb = cores, vc, stc
    if cores < (os.cpu_count() or 1):
        v2, st2 = run(2 * cores, nb)
        if v2 > 1.15 * vc:
            used, best, stb = 2 * cores, v2, st2
    return {"value": best, "unit": "Msamples/sec", "cores": used, "kind": "port",
            "host_cpu_count": os.cpu_count(), "affinity_and_quota_cpus": cores,
            "sample": "first %d spiral blocks (32x32 px, all %d spp) of the same %dx%d job, %d samples, %.1f s on %d threads" %
                      (nb, SPP, W, H, stb.samples, stb.seconds, used),
            "scaling_vs_one_thread": best / v1,
            "single_thread": {"value": v1, "unit": "Msamples/sec", "cores": 1,
                              "sample": "first %d spiral blocks at %d spp, %d samples, %.1f s" % (n1, SPP, st1.samples, st1.seconds)}}


LIVE_PASSES = (("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"),
               ("SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
               ("FETCH_SIZE",), ("WRITE_SIZE",))


def parse_counter_dirs(dirs):
    """{kernel base name: {counter: sum over its dispatches, "_dispatches": n}} of rocprofv3 --pmc output directories (csv)"""
    import csv
    import glob
    out = {}
    for d in dirs:
        seen = set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
                e = out.setdefault(k, {})
                name = r["Counter_Name"] + ("@2" if r["Counter_Name"] == "SQ_WAVE_CYCLES" and d.endswith("pass1") else "")   # (the wave cycles of the SQ_WAIT_ANY pass)
                e[name] = e.get(name, 0.0) + float(r["Counter_Value"])
                if d.endswith("pass0") and (k, r["Dispatch_Id"]) not in seen:
                    seen.add((k, r["Dispatch_Id"])); e["_dispatches"] = e.get("_dispatches", 0) + 1
    return out


def counters_to_entry(c):
    """HBM bytes and issue statistics of one kernel from its summed counters (per dispatch): FETCH_SIZE / WRITE_SIZE are in KiB,
    FETCH_SIZE under-reports a wide stream by 2x on gfx950 (MI355X_MICROARCH.md, section HBM); a wave64 VALU instruction
    occupies a SIMD-32 for 2 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    n = max(int(c.get("_dispatches", 1)), 1)
    rd, wr = c["FETCH_SIZE"] * 1024 * 2 / n, c["WRITE_SIZE"] * 1024 / n
    simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
    return {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
            "valu_issue_frac": c["SQ_INSTS_VALU"] * 2.0 / simd_cycles,
            "lane_util": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
            "wait_mem_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES@2"] if c.get("SQ_WAVE_CYCLES@2") else None}


def live_counters(argv_workload, kernel):
    """The PMC counters of the dominant kernel, collected IN THIS RUN: four rocprofv3 passes (each counter set in its own pass,
    --kernel-trace only, as MI355X_MICROARCH.md prescribes) over one untimed frame of the same workload in a child process.
    Returns the traffic.json-style entry of `kernel`, or None when rocprofv3 is not there or a pass fails (the caller then
    falls back to the committed profile of the same kernel sources)."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    tmp = tempfile.mkdtemp(prefix="miwave_pmc_", dir="/tmp")
    try:
        child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras",
                 "--no-live-counters"] + argv_workload
        dirs = []
        for i, ctrs in enumerate(LIVE_PASSES):
            d = os.path.join(tmp, "pass%d" % i); dirs.append(d)
            r = subprocess.run(["rocprofv3", "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            if r.returncode != 0:
                return None
        c = parse_counter_dirs(dirs).get(kernel)
        need = ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")
        if not c or any(k not in c for k in need):
            return None
        return counters_to_entry(c)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


GOLDEN_FRAMES = {      # the oracle's committed answers for the configurations AS CONFIGURED (tests/golden/make_golden_r3.py .. _r5.py)
    "c2": ("tests/golden/round3.json", "c2_full_1920x1080_512spp"),
    "c3": ("tests/golden/round4.json", "c3_full_1920x1080_1024spp"),
    "c4": ("tests/golden/round5.json", "c4_full_1920x1080_2048spp"),
    "c5": ("tests/golden/round4.json", "c5_full_1920x1080_512spp_spectral"),
}


def film_parity(film, which, samples=None, segments=None):
    """Ties the frame that was just TIMED to the oracle: sha256 of the float32 film on the device (one download, outside the timed
    region) against the digest the scalar oracle produced for this exact job (tests/golden/*.json, a committed fixture: nothing
    under oracle/ or /root/reference is read here), plus the sample / segment counts of the last frame. `match` is False when a
    kernel change moved a single bit of the film."""
    import hashlib
    path, key = GOLDEN_FRAMES[which]
    out = {"golden": "%s:%s" % (path, key), "film_sha256": None, "match": None}
    try:
        rec = json.load(open(os.path.join(ROOT, path)))[key]
        out["film_sha256"] = hashlib.sha256(film.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
        out["golden_sha256"] = rec["sha256"]
        out["match"] = out["film_sha256"] == rec["sha256"]
        if samples is not None:
            out["counts_match"] = (int(samples), int(segments)) == (int(rec["samples"]), int(rec["segments"]))
            out["match"] = bool(out["match"] and out["counts_match"])
    except Exception as e:
        out["error"] = repr(e)[:200]
    return out


def run_extras(api, scenes, film, C):
    """The other BASELINE configurations AS CONFIGURED, outside the timed headline, so that the driver's own bench line observes
    the tree kernels and the spectral variant at their configured sizes: configs[2] (material balls, 40 972 triangles) at 1024 spp,
    configs[3] class (0.9 M-triangle interior, area light + environment map) at 2048 spp, configs[4] (scalar_spectral glass-block
    box) at 512 spp — full 1920x1080 frames, one warm-up frame and one timed frame each (wall clock around mi_render, film on the
    device; a separate context per scene, so the headline context's counters stay the headline's). ~25 s in all."""
    import torch
    out = {}

    def timed(tag, scene, sensor, spp, note, golden=None, plan=0):
        device = api.Device(0)                   # world == 1: the benchmark runs on GPU 0
        try:
            # The tree is built twice and the SECOND build is quoted: the first device work after the previous configuration's context was
            # torn down (a 34 GB sample log after the material balls) stalls ~250 ms in one of the builder's host -> device uploads —
            # whichever comes first, and only in this order of events (gpurun r5l / r5m: "vertex-normal upload 268.9 ms" of a 277 ms
            # set-up; the same build in any other order: set-up 5 - 7 ms, tools/build_times.py). Both times are in the record.
            device.upload(scene.desc())
            first_build_ms = device.counters().ms_bvh_build
            device.upload(scene.desc())
            bvh = device.counters()
            job = api.PathIntegrator().render_job(sensor)
            cfg = job.cfg
            cfg.film_on_device = 1; cfg.film_f64 = 0; cfg.film_mode = 0; cfg.profile = 1; cfg.plan = plan; cfg.samples_per_launch = int(cfg.spp)
            device.check(device.L.mi_set_stream(device.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            ms = []
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                device.check(device.L.mi_render(device.ctx, C.byref(cfg), C.c_void_p(film.data_ptr())))
                torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
            c = device.counters()
            out[tag] = {"workload": note, "spp": spp, "value": 1920.0 * 1080 * spp / (ms[-1] * 1e-3) / 1e6, "unit": "Msamples/sec",
                        "ms_per_frame": ms[-1], "ms_first_frame": ms[0], "ms_path_kernel": c.ms_path, "ms_film": c.ms_resolve,
                        "samples": int(c.samples), "segments_per_sample": c.segments / max(c.samples, 1),
                        "path_kernel": ("k_path_pooled" if getattr(c, "pooled", 0) else "k_path_phased" if c.path_kernel in (1, 3) else "k_path_resident") if c.plan == 2 else
                                       ("k_trace_stream + k_sort_hits + k_shade" if c.path_kernel == 2 else "k_trace<closest|any> + k_shade"),
                        "plan": int(c.plan), "tree_width": int(c.tree_width), "film_overlapped": bool(getattr(c, "film_overlapped", 0)), "job_chunk": int(getattr(c, "job_chunk", 0)), "job_chunks": int(getattr(c, "job_chunks", 0)),
                        "kernel_ms": {"trace": round(c.ms_trace_closest, 2), "sort": round(c.ms_trace_any, 2), "shade": round(c.ms_shade, 2)} if c.plan == 1 else None,
                        "log_bytes": int(c.log_bytes),
                        "bvh": {"builder": {0: "host binned SAH", 1: "device LBVH", 3: "device binned SAH (level sweep)"}.get(bvh.bvh_builder, "device" if bvh.bvh_on_device else "host binned SAH"), "build_ms": round(bvh.ms_bvh_build, 1), "tris": bvh.bvh_tris,
                                "nodes2": bvh.bvh_nodes, "nodes8": bvh.bvh8_nodes, "depth8": bvh.bvh8_depth, "ms_bvh4": round(bvh.ms_bvh4, 2), "ms_bvh8": round(bvh.ms_bvh8, 2),
                                "build_ms_first_in_this_context": round(first_build_ms, 1)}}
            if golden:                            # the frame just timed against the oracle's digest of this configuration
                out[tag]["parity"] = film_parity(film, golden, c.samples, c.segments)
        finally:
            device.close()

    try:
        scene, sensor = scenes.cornell_box(1920, 1080, 1024, diffuse_only=False, device=-1)
        timed("c3_matball_1024spp", scene, sensor, 1024, "BASELINE configs[2] (GGX conductor + bk7 dielectric balls, 40 972 triangles), 1920x1080 @ its 1024 spp", "c3")
        # the same geometry through plan 1 — SoA ray / hit / shadow queues in HBM, ballot + prefix-sum compaction, material-sorted shading: the
        # architecture north_star names, kept as the both-plans twin of every parity test — at 256 spp; its film is the oracle's as well
        # (tests/test_gpu_parity.py, test_gpu_configured.py::test_c3_window*), its time is what the queue traffic costs
        scene, sensor = scenes.cornell_box(1920, 1080, 256, diffuse_only=False, device=-1)
        timed("c3_matball_plan1_256spp", scene, sensor, 256, "BASELINE configs[2] geometry through the wavefront plan (plan 1: SoA queues in HBM, one kernel per stage), 1920x1080 @ 256 spp", plan=1)
        scene, sensor = scenes.interior_scene(1920, 1080, 2048, device=-1)
        timed("c4_interior_2048spp", scene, sensor, 2048, "BASELINE configs[3] class (911 362 triangles, area light + 1024x512 environment map), 1920x1080 @ its 2048 spp, one GPU", "c4")
        if os.path.exists(api.default_srgb_coeff()):
            api.set_variant("scalar_spectral"); api.set_srgb_model(api.default_srgb_coeff())
            try:
                scene, sensor = scenes.cornell_box(1920, 1080, 512, diffuse_only=True, glass_block=True, device=-1)
                timed("c5_spectral_glassblock_512spp", scene, sensor, 512, "BASELINE configs[4] (scalar_spectral Cornell box with a bk7 dielectric block), 1920x1080 @ its 512 spp", "c5")
            finally:
                api.set_variant("scalar_rgb")
    except Exception as e:                    # the headline line must not die with an extra
        out["error"] = repr(e)[:300]
    return out


def spawn_command(n, argv, port=None):
    """`python bench.py --gpus N` without a launcher around it: the command line that runs this file as N ranks of one node
    (torch.distributed.run, rendezvous on 127.0.0.1 — the container hostname may not resolve), same arguments."""
    import socket
    if port is None:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n, argv):
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MIW_BENCH_SPAWNED="1")
    return subprocess.call(spawn_command(n, argv), env=env)


FILM_KERNELS = {0: "k_film_blocks", 1: "k_film_groups", 2: "k_film_columns", 3: "k_film_quads", 4: "k_film_lanes"}


def film_kernel_name(log_record_bytes, film_kernel):
    """the block replay mi_render launched (mi_counters::film_kernel; csrc/miwave.hip chooses by the log format, the shard's tile count
    and the MIW_FILM_* switches)"""
    if log_record_bytes != 16:
        return "k_film_blocks"
    return FILM_KERNELS.get(int(film_kernel), "k_film_%d" % film_kernel)


def reduce_label(backend):
    return {"nccl": "RCCL", "gloo": "gloo (host)"}.get(backend, backend)


REDUCE_HOW = {0: "none (one context)", 1: "device add, rank order (contexts share a GPU, or RCCL unavailable)", 2: "RCCL ncclReduce (mi_film_reduce)"}


def native_reduce_main(args):
    """`--native-reduce`: the N-GPU frame from ONE process through the C++ host layer — Scene::build(devices) puts the scene on N
    contexts, SamplingIntegrator::render runs one host thread per context over its shard of the spiral blocks (global block-id -> seed
    table) and closes the frame with mi_film_reduce; the root's film comes down to the host once per step (inside the timed region:
    that is what Integrator::render returns). `--dry-ranks`: the plan only (CPU tier): devices, reduce route, nothing rendered."""
    n = max(args.gpus, 1)
    devices = [0] * n if args.share_gpu else list(range(n))
    if args.dry_ranks:
        print(json.dumps({"n_gpus": n, "ranks_seen": len(devices), "backend": None, "route": "one process, %d contexts" % n, "devices": devices,
                          "reduce": "mi_film_reduce: " + (REDUCE_HOW[0] if n == 1 else REDUCE_HOW[1] if args.share_gpu else REDUCE_HOW[2])}), flush=True)
        return
    import numpy as np
    from mitsuba2_amd import api, scenes
    W, H, SPP = args.width, args.height, args.spp
    if args.variant != "scalar_rgb":
        api.set_variant(args.variant); api.set_srgb_model(api.default_srgb_coeff())
    if args.scene == "interior":
        scene, sensor = scenes.interior_scene(W, H, SPP, device=-1)
    elif args.scene == "glassblock":
        scene, sensor = scenes.cornell_box(W, H, SPP, diffuse_only=True, glass_block=True, device=-1)
    else:
        scene, sensor = scenes.cornell_box(W, H, SPP, diffuse_only=(args.scene == "cornell"), ball_level=args.tess, device=-1)
    scene.build(devices if n > 1 else devices[0], args.bvh_quality)
    integ = (api.DirectIntegrator if args.integrator == "direct" else api.PathIntegrator)()
    for _ in range(args.warmup):
        integ.render(scene, sensor)
    t0 = time.perf_counter()
    samples = segments = 0; ms_path = 0.0
    for _ in range(args.steps):
        assert integ.render(scene, sensor) is True
        c = integ.counters()
        samples += c.samples; segments += c.segments; ms_path += c.ms_path
    elapsed = time.perf_counter() - t0
    film = sensor.film.data((H, W, 5))
    value = float(W) * H * SPP * args.steps / elapsed / 1e6
    s_bar = segments / max(samples, 1)
    alg = 280.0 * segments + B_SPLAT * samples
    print(json.dumps({
        "metric": "Msamples/sec (whole node), 1080p/512spp path integrator", "value": value, "unit": "Msamples/sec",
        "n_gpus": n, "ranks_seen": scene.device_count(), "route": "one process: one context + one host thread per GPU (Scene::build(devices)), film reduced by mi_film_reduce",
        "reduce_how": REDUCE_HOW.get(integ.last_reduce(), str(integ.last_reduce())), "devices": devices,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "variant": args.variant, "integrator": args.integrator, "shard": "tiles",
        "config": {"workload": "%s, %dx%d @ %d spp, %s integrator" % (args.scene, W, H, SPP, args.integrator),
                   "parallelism": "tile-shard x%d in one process + mi_film_reduce" % n if n > 1 else "single GPU",
                   "film": "downloaded to the host once per step (inside the timed region)"},
        # the slowest context's path-kernel time per step (mi_counters::ms_path is the maximum over the contexts) against its share of the bytes
        "roofline": {"bound": None, "kernel": "path kernel (slowest context)", "achieved": (alg / n) / (ms_path * 1e-3) / 1e9 if ms_path > 0 else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (alg / n) / (ms_path * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_path > 0 else None,
                     "traffic": None, "segments_per_sample": s_bar},
        "cpu_baseline": None, "film_mean_y": float(np.asarray(film)[..., 1].mean()),
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true", help="do not collect the dominant kernel's PMC counters in this run (four "
                    "rocprofv3 passes over one extra frame in a child process, ~40 s; N = 1 only); roofline.traffic / .measured then come "
                    "from profiles/traffic.json while the kernel sources still hash to what was profiled")
    ap.add_argument("--no-extras", action="store_true", help="skip the `extras` block (the other BASELINE configurations at a few "
                    "spp each, after the timed headline; only the default N = 1 Cornell run carries it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for "
                    "exercising the multi-rank path on a box with fewer GPUs than ranks, together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--native-reduce", action="store_true", help="N > 1 from ONE process: the scene is built on N contexts (Scene::build(devices): one context + one "
                    "host thread per GPU), SamplingIntegrator::render shards the spiral blocks over them and mi_film_reduce sums the partial films "
                    "(RCCL ncclReduce between distinct GPUs, a rank-ordered device add when contexts share one) - the route next to the "
                    "torch.distributed one (one process per GPU). The JSON line carries ranks_seen and reduce_how")
    ap.add_argument("--dry-ranks", action="store_true", help="testing only (CPU tier): every rank joins the process group, takes part in one "
                    "all-reduce and rank 0 prints {n_gpus, ranks_seen}; nothing is rendered")
    ap.add_argument("--shard-of", type=int, default=0, help="testing only (N = 1): render one shard of this many — what one rank "
                    "of an N-GPU run executes (--shard-index says which, default 0); `value` then counts only that shard's samples")
    ap.add_argument("--shard-index", type=int, default=0, help="with --shard-of N: the rank whose shard is rendered (0 .. N - 1); "
                    "tools/shard_table.py renders all of them and reports the slowest, which is what an N-GPU frame lasts")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--film-mode", type=int, default=0, help="0 auto, 1 sample log + ordered gather, 2 float64 atomics")
    ap.add_argument("--scene", default="cornell", choices=["cornell", "matball", "interior", "glassblock"],
                    help="cornell = BASELINE configs[1] (diffuse Cornell box); matball = configs[2] (GGX rough conductor + "
                         "dielectric balls, 41k triangles; quoted at 1024 spp); interior = configs[3] class (0.9 M triangles, area "
                         "light + environment map); glassblock = configs[4]'s geometry (Cornell box with a dielectric block; with "
                         "--variant scalar_spectral)")
    ap.add_argument("--tess", type=int, default=5, help="matball: icosphere subdivision level of the two balls (5 = 40 972 triangles, "
                    "the config; 0..4 = 52 / 172 / 652 / 2 572 / 10 252: the triangle-count series of DESIGN.md)")
    ap.add_argument("--variant", default="scalar_rgb", choices=["scalar_rgb", "scalar_spectral"],
                    help="scalar_spectral = BASELINE configs[4] (4 wavelengths per sample; needs mitsuba2_amd/data/srgb.coeff or "
                         "MIWAVE_SRGB_COEFF = the reference's data/srgb.coeff)")
    ap.add_argument("--bvh-quality", type=int, default=0, help="0 = the binned-SAH tree built on the device (csrc/sah_device.h: the default; + 64 = MI_BVH_RADIX_TREE: the radix tree of rounds 2 - 3), 1 = the same tree built by the host recursion")
    ap.add_argument("--shard", default="auto", choices=["auto", "tiles", "passes"],
                    help="how N ranks split the frame. tiles (the north star's partition; what auto picks, always): spiral blocks dealt "
                         "round-robin, every rank renders all spp of its pixels; the N-GPU film equals the 1-GPU film. passes (explicit "
                         "request only): the reference's own samples_per_pass = spp / N run (integrator.cpp:75-86, spiral.cpp:41) with "
                         "pass r rendered by rank r — every rank keeps all pixels, but the film is the one scalar_rgb produces for that "
                         "samples_per_pass: another set of random numbers than the 1-GPU job (DESIGN.md section 7)")
    ap.add_argument("--integrator", default="path", choices=["path", "direct"],
                    help="path = the headline (BASELINE metric); direct = src/integrators/direct.cpp on the same device loop")
    ap.add_argument("--plan", type=int, default=0, help="0 auto, 1 wavefront (HBM queues), 2 resident (registers + LDS)")
    ap.add_argument("--samples-per-launch", type=int, default=-1,
                    help="resident plan: samples each pixel advances per launch (-1 = all spp in one launch, 0 = library default)")
    args = ap.parse_args()

    if args.native_reduce:
        return native_reduce_main(args)
    # N ranks: under a launcher (the driver's torch.distributed.run: WORLD_SIZE is set) this process is one of them; alone with
    # --gpus N > 1 it becomes the launcher of N ranks of itself
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's rank count is what runs (n_gpus = %d)" % (args.gpus, world, world), file=sys.stderr)

    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.dry_ranks:
        seen = 1
        if world > 1:
            t = torch.ones(1, dtype=torch.int64)
            if args.backend == "nccl":
                torch.cuda.set_device(0 if args.share_gpu else local_rank); t = t.cuda()
            dist.all_reduce(t); seen = int(t.item())
            dist.barrier(); dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks_seen": seen, "backend": args.backend if world > 1 else None,
                              "reduce": reduce_label(args.backend) if world > 1 else None}), flush=True)
        return
    import numpy as np
    from mitsuba2_amd import api, scenes, _capi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: mitsuba2_amd has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.barrier()          # creates the RCCL communicator now, outside the timed region (also with --warmup 0)

    W, H, SPP = args.width, args.height, args.spp
    if args.variant != "scalar_rgb":
        api.set_variant(args.variant)
        if not os.environ.get("MIWAVE_SRGB_COEFF"):
            api.set_srgb_model(api.default_srgb_coeff())   # mitsuba2_amd/data/srgb.coeff: the table the reference's build generates
    if args.scene == "interior":     # BASELINE configs[3] class: ~0.9 M triangles, area light + environment map
        scene, sensor = scenes.interior_scene(W, H, SPP, device=-1)
    elif args.scene == "glassblock":
        scene, sensor = scenes.cornell_box(W, H, SPP, diffuse_only=True, glass_block=True, device=-1)
    else:
        scene, sensor = scenes.cornell_box(W, H, SPP, diffuse_only=(args.scene == "cornell"), ball_level=args.tess, device=-1)
    dev = api.Device(local_rank)
    dev.upload(scene.desc(), bvh_quality=args.bvh_quality)   # scene + BVH resident before timing
    bvh = dev.counters()
    make_integrator = api.DirectIntegrator if args.integrator == "direct" else api.PathIntegrator
    parts = world if world > 1 else max(args.shard_of, 1)          # ranks, or the rank count --shard-of stands for
    from mitsuba2_amd import dist as mdist
    shard = mdist.choose_shard(args.shard, parts, W, H, SPP)
    if shard == "passes":
        integ, job = mdist.pass_job(make_integrator, sensor, rank, parts, SPP)   # pass r carries block ids (N - 1 - r) * block_count + counter
    else:
        integ = make_integrator()
        integ.set_shard(rank, world)
        if args.shard_of > 1 and world == 1:
            if not 0 <= args.shard_index < args.shard_of:
                raise SystemExit("bench.py: --shard-index must lie in 0 .. --shard-of - 1")
            integ.set_shard(args.shard_index, args.shard_of)
        job = integ.render_job(sensor)
    cfg = job.cfg
    cfg.film_on_device = 1; cfg.film_f64 = 0; cfg.film_mode = args.film_mode; cfg.profile = 0 if args.no_profile else 1
    cfg.plan = args.plan; cfg.samples_per_launch = int(cfg.spp) if args.samples_per_launch < 0 else args.samples_per_launch
    film = torch.zeros(H * W * 5, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    dev.check(dev.L.mi_set_stream(dev.ctx, C.c_void_p(stream.cuda_stream)))

    def step():
        st = dev.L.mi_render(dev.ctx, C.byref(cfg), C.c_void_p(film.data_ptr()))
        dev.check(st)
        if world > 1:
            dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    agg = dict(ms_shade=0.0, ms_tc=0.0, ms_ta=0.0, ms_film=0.0, ms_init=0.0, n_shade=0, n_tc=0, n_ta=0, segments=0, samples=0,
               shadow=0, iters=0, ms_path=0.0, n_path=0, ms_fb=0.0, ms_fm=0.0, ms_fp=0.0, n_film=0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        c = dev.counters()
        agg["ms_shade"] += c.ms_shade; agg["ms_tc"] += c.ms_trace_closest; agg["ms_ta"] += c.ms_trace_any
        agg["ms_film"] += c.ms_resolve; agg["ms_init"] += c.ms_init
        agg["n_shade"] += c.n_shade; agg["n_tc"] += c.n_trace_closest; agg["n_ta"] += c.n_trace_any
        agg["segments"] += c.segments; agg["samples"] += c.samples; agg["shadow"] += c.shadow_rays; agg["iters"] += c.iterations
        agg["ms_path"] += c.ms_path; agg["n_path"] += c.n_path; agg["ms_fb"] += c.ms_film_blocks; agg["ms_fm"] += c.ms_film_merge; agg["ms_fp"] += c.ms_film_pack
        agg["n_film"] += 1
    sync()
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(t); ranks_seen = int(t.item())

    if rank == 0:
        total_samples = float(W) * H * SPP * args.steps
        if args.shard_of > 1 and world == 1:
            total_samples = float(agg["samples"])
        value = total_samples / elapsed / 1e6
        s_bar = agg["segments"] / max(agg["samples"], 1)
        hc = dev.counters()                      # the headline frames' counters (nothing rendered after them may leak into the line)
        pk = hc.path_kernel
        path_kernel = "k_path_phased" if pk in (1, 3) else "k_path_resident"
        tc_name, ta_name = ("k_trace_stream", "k_sort_hits") if pk == 2 else ("k_trace<closest>", "k_trace<any>")
        # dominant kernel by summed HIP-event time (rank 0's shard)
        kernels = {
            "k_shade": (agg["ms_shade"], agg["n_shade"], B_SHADE * agg["segments"] + B_SPLAT * agg["samples"]),
            # plan 1 over a tree: one persistent stream kernel walks the E and the S rays (+ k_sort_hits: 4 B of key per segment)
            tc_name: (agg["ms_tc"], agg["n_tc"], B_TRACE_CLOSEST * agg["segments"] + (B_TRACE_ANY * agg["shadow"] if pk == 2 else 0.0)),
            ta_name: (agg["ms_ta"], agg["n_ta"], 8.0 * agg["segments"] if pk == 2 else B_TRACE_ANY * agg["shadow"]),
            # resident plan: the whole pipeline's algorithmic bytes (280 B/segment + 320 B/sample) belong to one kernel
            path_kernel: (agg["ms_path"], agg["n_path"], 280.0 * agg["segments"] + B_SPLAT * agg["samples"]),
            # ordered film replay (device/film_kernels.h): k_film_lanes over the tile-interleaved 16-byte log (k_film_quads for shards of
            # fewer than 448 tiles), k_film_blocks over the 24-byte position log of filters without phase classes
            film_kernel_name(hc.log_record_bytes, hc.film_kernel): (agg["ms_fb"], agg["n_film"], float(hc.log_record_bytes) * agg["samples"]),
        }
        roofline = None
        if not args.no_profile and (agg["n_shade"] or agg["n_path"]):
            name = max(kernels, key=lambda k: kernels[k][0])
            ms, n, alg_bytes = kernels[name]
            achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            b_alg = 280.0 * s_bar + 320.0
            # What the counters say about that kernel: collected in this run (live_counters: rocprofv3 PMC passes over one more frame
            # in a child process), else the committed profiles/traffic.json entry of this exact workload — used only while the
            # kernel sources still hash to what the profile was taken on (kernel_src_sha16); otherwise the fields stay null.
            traffic = None; measured = None
            entry = None; source = None
            if world == 1 and not args.no_live_counters and args.shard_of <= 1 and not os.environ.get("MIW_BENCH_NO_LIVE"):
                workload = ["--width", str(W), "--height", str(H), "--spp", str(SPP), "--scene", args.scene, "--tess", str(args.tess),
                            "--variant", args.variant, "--bvh-quality", str(args.bvh_quality), "--integrator", args.integrator,
                            "--plan", str(args.plan), "--film-mode", str(args.film_mode), "--samples-per-launch", str(args.samples_per_launch)]
                entry = live_counters(workload, name)
                if entry:
                    source = "live: 4 rocprofv3 --pmc passes over one frame of this workload, in this run"
            if entry is None:
                try:
                    key = "%s/%s/%dx%d@%d/plan%d/film%d/launch%d" % (args.variant, args.scene, W, H, SPP, hc.plan,
                                                                     hc.film_mode, cfg.samples_per_launch)
                    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                    if world == 1 and table.get(key, {}).get("kernel_src_sha16") == kernel_src_sha16():
                        entry = table.get(key, {}).get(name)
                        source = "committed: " + str(table[key].get("source"))
                except Exception:
                    entry = None
            if entry:
                traffic = entry["hbm_bytes_per_launch"]
                measured = {"source": source, "hbm_bytes_per_launch": traffic,
                            # the real bound next to the decreed one: HBM bytes the kernel really moves / its time / 8 TB/s,
                            # the share of SIMD issue cycles that carried a VALU instruction, the lanes those instructions used
                            "hbm_measured_frac": traffic / (ms / max(n, 1) * 1e-3) / (HBM_PEAK_GBS * 1e9),
                            "valu_issue_frac": entry.get("valu_issue_frac"), "lane_use": entry.get("lane_util"),
                            "wait_mem_frac": entry.get("wait_mem_frac")}
            # `bound` names what limits the kernel on the silicon: the vector ALUs (issue slots x lanes per instruction, from the
            # committed PMC passes of these exact kernel sources) unless the measured HBM traffic is the larger fraction;
            # achieved / peak / frac stay the yardstick north_star decrees (ALGORITHMIC queue + splat bytes of SURVEY.md 8d
            # over the kernel time against 8 TB/s) — a path-tracing kernel that keeps its state in registers moves a few
            # per cent of those bytes for real (`traffic`), so that fraction is a throughput scale, not a bandwidth claim.
            bound, bound_frac = None, None          # unmeasured (no PMC counters for this run): say so rather than guess
            if measured and measured.get("valu_issue_frac") is not None and measured.get("lane_use") is not None:
                valu = measured["valu_issue_frac"] * measured["lane_use"]
                bound, bound_frac = ("valu", valu) if valu >= measured["hbm_measured_frac"] else ("hbm", measured["hbm_measured_frac"])
            roofline = {
                "bound": bound, "bound_frac": bound_frac,
                "yardstick": "hbm, algorithmic bytes (280 B/segment + 320 B/sample, SURVEY.md section 8d) / kernel time / 8 TB/s",
                "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "measured": measured,
                "launches": n, "avg_launch_ms": ms / max(n, 1), "alg_bytes_per_launch": alg_bytes / max(n, 1),
                "kernel_ms": dict({k: round(v[0], 3) for k, v in kernels.items() if v[1]},
                                  k_film_merge=round(agg["ms_fm"], 3), k_init=round(agg["ms_init"], 3)),
                "log_bytes": hc.log_bytes, "log_record_bytes": hc.log_record_bytes,
                # round 6: the film replay queued beside the path kernel (one launch per group of 64 tiles on a second stream): its kernel_ms then overlap the path kernel's
                "film_overlapped": bool(getattr(hc, "film_overlapped", 0)), "film_groups": int(getattr(hc, "film_groups", 0)),
                # round 6: the pixels' sample streams cut into halving chunks (smallest: job_chunk samples; job_chunks per pixel) drawn chunk-major from one queue
                "job_chunk": int(getattr(hc, "job_chunk", 0)), "job_chunks": int(getattr(hc, "job_chunks", 0)),
                "segments_per_sample": s_bar,
                "pipeline_alg_bytes_per_sample": b_alg,
                "pipeline_frac": (value / world) * 1e6 * b_alg / (HBM_PEAK_GBS * 1e9),
            }
        cpu = None
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            O = oracle_py.load(args.variant)
            cpu = cpu_baseline(O, np, scene, make_integrator, sensor, W, H, SPP)
        # parity of the frame that was timed: the device film of the LAST timed step against the oracle's committed digest of this
        # exact job (the default headline workload only; an N-rank film differs from it by <= 1 ulp at block borders and is not hashed)
        parity = None
        headline = (world == 1 and args.scene == "cornell" and args.variant == "scalar_rgb" and args.shard_of <= 1
                    and args.integrator == "path" and (W, H, SPP) == (1920, 1080, 512) and shard != "passes")
        if headline and args.steps > 0:
            parity = film_parity(film, "c2", hc.samples, hc.segments)
        extras = None
        if (not args.no_extras and world == 1 and args.scene == "cornell" and args.variant == "scalar_rgb" and args.shard_of <= 1
                and args.integrator == "path" and (W, H, SPP) == (1920, 1080, 512)):
            extras = run_extras(api, scenes, film, C)
        out = {
            "metric": "Msamples/sec (whole node), 1080p/512spp path integrator", "value": value, "unit": "Msamples/sec",
            "n_gpus": world, "ranks_seen": ranks_seen, "route": "one process per GPU (torch.distributed)" if world > 1 else "one process, one GPU",
            "reduce_how": ("torch.distributed.reduce(sum) over " + reduce_label(args.backend)) if world > 1 else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "variant": args.variant, "integrator": args.integrator, "shard": shard if world > 1 or args.shard_of > 1 else None,
            "shard_index": args.shard_index if args.shard_of > 1 and world == 1 else None,
            "config": {"workload": ("Cornell box (32 triangles), %dx%d @ %d spp, diffuse-only BSDFs, path integrator "
                                    "max_depth=-1 rr_depth=5, gaussian rfilter, independent sampler seed 0" if args.scene == "cornell" else
                                    "Cornell box with a bk7 dielectric block (34 triangles), %dx%d @ %d spp, path integrator max_depth=-1 "
                                    "rr_depth=5, gaussian rfilter, independent sampler seed 0" if args.scene == "glassblock" else
                                    "procedural interior (911 362 triangles: displaced wall grids + 200 icospheres, diffuse / GGX / "
                                    "Beckmann conductors / dielectric), area light + 1024x512 environment map, %dx%d @ %d spp, path "
                                    "integrator max_depth=-1 rr_depth=5" if args.scene == "interior" else
                                    "material balls in the Cornell box (GGX rough conductor + bk7 dielectric icospheres, " + str(bvh.bvh_tris) +
                                    " triangles, shading normals), %dx%d @ %d spp, path integrator max_depth=-1 rr_depth=5, "
                                    "gaussian rfilter, independent sampler seed 0") % (W, H, SPP),
                       "bvh": {"builder": {0: "host binned SAH", 1: "device LBVH", 3: "device binned SAH (level sweep)"}.get(bvh.bvh_builder, "device" if bvh.bvh_on_device else "host binned SAH"), "build_ms": round(bvh.ms_bvh_build, 3),
                               "nodes": bvh.bvh_nodes, "tris": bvh.bvh_tris, "depth": bvh.bvh_depth},
                       "parallelism": ("%s-shard x%d + %s film reduce" % ("tile" if shard == "tiles" else "pass (samples_per_pass = spp / %d)" % world, world, reduce_label(args.backend))) if world > 1 else "single GPU",
                       "plan": {1: "wavefront: SoA queues in HBM, one kernel per stage" + (" (persistent stream walk kernel with dynamic ray fetch)" if pk == 2 else ""), 2: "resident: path state in registers, geometry in LDS"
                                if path_kernel == "k_path_resident" else "resident, wave-level phase machine: path + walk state in registers, "
                                "per-lane LDS stack, nodes / triangles through L1 / L2" + (" (%d-wide quantised tree)" % hc.tree_width if hc.tree_width else "")}[hc.plan],
                       "film": {1: "sample log (%d B per sample) + ordered float32 gather (bit-identical to scalar_rgb order)" % hc.log_record_bytes, 2: "float64 atomics"}[hc.film_mode]},
            "parity": parity, "roofline": roofline, "cpu_baseline": cpu, "extras": extras,
        }
        if args.integrator == "direct":
            out["config"]["workload"] = out["config"]["workload"].replace("path integrator max_depth=-1 rr_depth=5", "direct integrator shading_samples=1")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
