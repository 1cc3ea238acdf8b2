#!/usr/bin/env python3
"""bench.py — UNet denoise steps/sec on MI355X for the reference's per-view paint loop.

A "step" is one iteration of the denoise loop of StableDiffusion.img2img_step
(src/stable_diffusion_depth.py:331-514): CFG-batched UNet evaluation (batch 2: [uncond, text]) on
[latents ; depth] followed by the CFG combine + PNDM/PLMS update.  N=1 workload = BASELINE.json configs[1]:
one view @768^2 (latent 96^2), 50 scheduler steps, SD2-depth architecture, fp16 MFMA kernels, seeded
random-init weights and synthetic text/depth inputs (no network for checkpoints or datasets).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N>1: one process per GPU; the 6-8 views of a mesh shard one per rank (weak scaling: every rank denoises its
own view, no data-path collective inside the denoise loop); value = total steps of all ranks / max time.
The line also carries `sec_per_mesh`: ONE measured ConTEXTure.paint (6 views over the job's ranks, both
all-reduces inside).  `--mode mesh` makes that the timed quantity (a step = one mesh; strong scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--latent", type=int, default=96, help="latent side (96 = 768^2 image; 64 = 512^2)")
    p.add_argument("--guidance", type=float, default=10.0)
    p.add_argument("--cpu-baseline", type=int, default=1)
    p.add_argument("--cpu-latent", type=int, default=0, help="latent side of the CPU sample (0 = auto)")
    p.add_argument("--vae", type=int, default=1, help="also time the once-per-view VAE decode (outside the step loop)")
    p.add_argument("--two-views", type=int, default=1, help="also measure 2 views in flight on one GPU (outside the timed region)")
    p.add_argument("--mode", choices=("steps", "mesh"), default="steps",
                   help="steps: UNet denoise steps/s (the driver's line).  mesh: a step is one whole ConTEXTure.paint of a 6-view mesh "
                        "(views sharded over the ranks, both all-reduces inside): measured sec/mesh, strong scaling")
    p.add_argument("--mesh", type=int, default=1, help="steps mode: also MEASURE sec/mesh with one ConTEXTure.paint (outside the timed region)")
    p.add_argument("--mesh-path", default="shapes/nascar.obj")
    p.add_argument("--mesh-views", type=int, default=6)
    p.add_argument("--in-flight", type=int, default=3, help="views a rank keeps in its denoise loop at once (optim.views_in_flight)")
    p.add_argument("--per-eval", type=int, default=6, help="mesh leg: views a rank denoises in lockstep as ONE UNet evaluation of batch 2 x this "
                                                            "(optim.views_per_eval; 0 / 1: --in-flight streams of batch 2 instead)")
    return p.parse_args()


def make_painter(a, dev, unet):
    """ConTEXTure over the bench's UNet engine: bundled mesh, Zero123PlusDataset views 1..6, 1200^2 render, latent a.latent,
    50 PLMS steps, VAE decode, view weights, UV scatter, atlas merge (BASELINE metric, first half: sec/mesh)."""
    from contexture_nerf_amd import config as CFG
    from contexture_nerf_amd.trainer import ConTEXTure
    from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion
    cfg = CFG.TrainConfig()
    cfg.guide.text = "a photo of a car"
    cfg.guide.shape_path = a.mesh_path
    cfg.guide.guidance_scale = a.guidance
    cfg.guide.sd_image_size = a.latent * 8
    cfg.guide.num_inference_steps = 50
    cfg.optim.views_in_flight = a.in_flight
    cfg.optim.views_per_eval = a.per_eval if a.per_eval > 1 else 0
    sd = StableDiffusion(dev, unet=unet)
    tr = ConTEXTure(cfg, device=dev, diffusion=sd)
    tr.train_views = tr.train_views[1:1 + a.mesh_views]
    tr.text_z = sd.get_text_embeds([cfg.guide.text])
    return tr


def timed_paints(tr, n, warm, dist, dev):
    """n ConTEXTure.paint calls bracketed by barrier + synchronize; -> max-over-ranks seconds, atlas coverage."""
    for _ in range(warm):
        tr.paint()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        atlas, cov = tr.paint()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    assert torch.isfinite(atlas).all()
    return el, float((cov > 0).float().mean())


def cpu_baseline(latent, budget_s=60.0, forced=0):
    """The oracle's fp32 PyTorch UNet (the reference's CPU behaviour: autocast is a no-op on CPU,
    stable_diffusion_depth.py:330) timed on this box's host cores on ONE CFG-batched step."""
    from oracle import unet_ref
    # a 1-GPU box owns a 16-core share of the host (more threads only oversubscribe it)
    cores = int(os.environ.get("CTX_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    net = unet_ref.UNet2DConditionModelRef(unet_ref.SD2_DEPTH).eval()
    build_s = time.perf_counter() - t0
    ctx = torch.randn(2, 77, 1024)

    def run(side):
        x = torch.randn(2, 5, side, side)
        with torch.no_grad():
            t = time.perf_counter()
            net(x, torch.tensor(501.0), ctx)
            return time.perf_counter() - t
    probe = run(32)                                           # 0.36 TFLOP: also warms the thread pool
    fl = {s: 2 * unet_ref.count_flops(unet_ref.SD2_DEPTH, s, s)['total'] for s in (32, 64, 96)}
    side = forced or latent
    if not forced:
        while side > 32 and probe * fl.get(side, fl[96]) / fl[32] > budget_s:
            side = {96: 64, 64: 32}.get(side, 32)
    dt = run(side) if side != 32 else min(probe, run(32))
    scale = fl[side] / fl.get(latent, fl[96])                 # FLOP-ratio extrapolation when the sample is smaller
    step_s = dt / scale
    return {"value": round(1.0 / step_s, 5), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"1 CFG-batched UNet step (batch 2) at latent {side}^2, fp32 torch CPU oracle, {cores} threads, "
                      f"{dt:.2f} s measured" + ("" if side == latent else f"; scaled to latent {latent}^2 by FLOP ratio {1 / scale:.2f}x")
                      + f" (model build {build_s:.1f} s not counted)"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local = local % max(torch.cuda.device_count(), 1)         # rehearsal: several ranks may share one card (gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        dist.init_process_group(os.environ.get("CTX_BENCH_BACKEND", "nccl"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("CTX_BENCH_TIMEOUT_S", "300"))))

    from contexture_nerf_amd import _lib as L
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from contexture_nerf_amd.scheduler import PNDMScheduler
    lib = L.load()
    L.check(lib.ctx_device_check())

    S = a.latent
    unet = UNet2DConditionModel(device=dev, seed=0)           # SD2-depth architecture, 866 M params, random init

    def make_view(net, seed):
        """One view's denoise loop state: own latents / depth / text embeddings / scheduler; returns its step function."""
        g = torch.Generator(device=dev).manual_seed(seed)
        text_z = torch.randn(2, 77, 1024, generator=g, device=dev)
        depth = torch.rand(1, 1, S, S, generator=g, device=dev) * 2 - 1
        depth2 = torch.cat([depth] * 2)
        sched = PNDMScheduler()
        state = {"i": 0, "lat": None}

        def new_view():
            sched.set_timesteps(50)
            state["i"] = 0
            state["lat"] = torch.randn(1, 4, S, S, generator=g, device=dev)

        def step():
            if state["lat"] is None or state["i"] >= len(sched.timesteps):
                new_view()
            t = int(sched.timesteps[state["i"]])
            lat = state["lat"]
            x = torch.cat([torch.cat([lat] * 2), depth2], dim=1)                      # [2,5,S,S]
            eps = net(x, float(t), encoder_hidden_states=text_z)['sample']           # K14
            state["lat"] = sched.step_cfg(eps, a.guidance, t, lat)['prev_sample']   # K15+K16 fused
            state["i"] += 1
        return step, state

    if a.mode == "mesh":
        # strong scaling: ONE mesh, its views sharded over the ranks, measured wall time per mesh
        tr = make_painter(a, dev, unet)
        el, cover = timed_paints(tr, a.steps, max(1, a.warmup), dist, dev)
        if rank == 0:
            print(json.dumps({
                "metric": "sec/mesh full texture (6 views, 50 PLMS steps)", "value": round(el / a.steps, 4), "unit": "s/mesh",
                "n_gpus": world, "steps": a.steps, "warmup": max(1, a.warmup), "ms_per_step": round(el / a.steps * 1e3, 2),
                "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[2]/[1] shapes: {a.mesh_path}, {a.mesh_views} views (Zero123PlusDataset 1..{a.mesh_views}) "
                                       f"@1200^2 render, SD2-depth fp16 at latent {S}^2, 50 PLMS steps (51 UNet evals) per view, VAE decode, "
                                       "view weights + UV scatter, all-reduce(MAX) [F] + all-reduce(SUM) [4,1024,1024]; random-init weights",
                           "views": a.mesh_views, "views_in_flight": a.in_flight, "views_per_eval": a.per_eval, "parallelism": f"view-shard x{world}"},
                "atlas_coverage": round(cover, 4)}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    step, state = make_view(unet, 1234 + rank)               # each rank = its own view

    for _ in range(a.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(state["lat"]).all(), "non-finite latents"

    # ---- live per-kernel timing (outside the timed region): dominant kernel = fp16 MFMA GEMM / implicit conv ------
    fl = unet.flops(2, S, S, 77)
    L.check(lib.ctx_profile_begin())
    step()
    torch.cuda.synchronize()
    prof = {}
    for k, name in ((0, "gemm_conv"), (1, "attention")):
        ms, n = C.c_double(), C.c_int64()
        L.check(lib.ctx_profile_end(k, C.byref(ms), C.byref(n)))
        prof[name] = (ms.value, n.value)
    gemm_ms, gemm_n = prof["gemm_conv"]
    att_ms, att_n = prof["attention"]
    gemm_fl = fl["gemm_conv"][1]
    achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    total_fl = sum(v[1] for v in fl.values())

    # once-per-view tail of img2img_step: VAE decode of the denoised latents (stable_diffusion_depth.py:567), timed outside
    # the step loop
    vae_ms, vae_tflop = None, None
    if a.vae:
        from contexture_nerf_amd.vae import AutoencoderKL
        vae = AutoencoderKL(device=dev, seed=0)
        zlat = state["lat"] / 0.18215
        vae.decode(zlat); torch.cuda.synchronize()
        tv = time.perf_counter()
        img = vae.decode(zlat).sample
        torch.cuda.synchronize()
        vae_ms = (time.perf_counter() - tv) * 1e3
        assert torch.isfinite(img).all()
        vae_tflop = vae.flops() / 1e12

    # several views of the mesh in flight on one GPU (n HIP streams, n engines over ONE weight blob): the deep UNet levels
    # do not fill the chip, so a rank that owns several views paints them 2-3 at a time.  Reported beside `value`, which
    # stays the single-view figure of BASELINE configs[1].
    two = None
    if a.two_views:
        two = {"note": "n views concurrently on n HIP streams, engines share one weight blob (UNet2DConditionModel.clone_shared)"}
        views = [(step, state)]
        for n in (2,):      # 3 in flight is measured at trainer level (tools/bench_mesh.py -> profiles/r01_mesh_bench.json: 2.42 s / mesh)
            while len(views) < n:
                views.append(make_view(unet.clone_shared(), 4321 + 17 * len(views) + rank))
            sts = [torch.cuda.Stream() for _ in range(n)]
            torch.cuda.synchronize()

            def group(k):
                for _ in range(k):
                    for v in range(n):
                        with torch.cuda.stream(sts[v]):
                            views[v][0]()
            group(max(5, a.warmup)); torch.cuda.synchronize()     # a fresh engine clone sizes its workspace in its first rounds
            t2 = time.perf_counter()
            group(a.steps); torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            assert all(torch.isfinite(v[1]["lat"]).all() for v in views[:n])
            two[str(n)] = {"steps_per_s": round(n * a.steps / dt2, 3), "ms_per_step_per_view": round(dt2 / (n * a.steps) * 1e3, 3)}

    # multi-GPU exchange step of the path (once per mesh, not per denoise step): atlas all-reduce, timed separately
    atlas_ms = None
    if dist is not None:
        atlas = torch.zeros(4, 1024, 1024, device=dev)
        dist.all_reduce(atlas); torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.all_reduce(atlas); torch.cuda.synchronize()
        atlas_ms = (time.perf_counter() - t1) * 1e3

    # measured sec/mesh (BASELINE metric, first half): one whole ConTEXTure.paint over this job's ranks, outside the timed region
    mesh_s, mesh_cover = None, None
    mesh_hung = False
    if a.mesh:
        # The leg runs in a worker thread under a deadline: an exception is reported in the line, and a rank that never comes back
        # (a peer died before a collective) cannot take the steps/s line with it — rank 0 prints without the figure and every
        # rank leaves through os._exit below.
        import threading
        box = {}

        def _mesh_leg():
            try:
                torch.cuda.set_device(dev)
                tr = make_painter(a, dev, unet)
                box["res"] = timed_paints(tr, 1, 1, dist, dev)
                del tr
            except Exception as e:                          # never lose the steps/s line to the mesh leg
                box["res"] = (None, f"failed: {e}")
                import traceback
                traceback.print_exc()

        th = threading.Thread(target=_mesh_leg, daemon=True)
        th.start()
        th.join(float(os.environ.get("CTX_BENCH_MESH_TIMEOUT_S", "180")))
        if th.is_alive():
            mesh_hung = True
            mesh_s, mesh_cover = None, "timed out (a rank did not return from the mesh leg)"
        else:
            mesh_s, mesh_cover = box.get("res", (None, "failed: no result"))

    # HBM-side bytes per launch of the dominant kernel family: PMC counters cannot be collected from inside this process,
    # so the figure is the committed result of tools/pmc_traffic.sh (same command line, same workload) when present
    traffic, traffic_src = None, None
    import glob
    tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))      # newest round's PMC pass
    tf = tfs[-1] if tfs else ""
    if tf:
        try:
            tj = json.load(open(tf))
            traffic = round(tj["families"]["gemm_conv"]["bytes_per_launch"])
            traffic_src = f"profiles/{os.path.basename(tf)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, per GEMM/conv launch)"
        except Exception:
            traffic = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        out = {
            "metric": "UNet denoise steps/sec (CFG-batched SD2-depth UNet eval + PLMS update)",
            "value": round(world * a.steps / elapsed, 4), "unit": "steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: 1 view @{S * 8}^2 (latent {S}^2), 50 PLMS steps (51 UNet evals), "
                                   "SD2-depth UNet fp16, CFG batch 2, guidance 10, random-init weights; "
                                   "mesh substitution napoleon.obj -> n/a for this stage (denoise loop only)",
                       "latent": S, "cfg_batch": 2, "ctx_len": 77, "views_per_rank": 1, "parallelism": f"view-shard x{world}"},
            "tflops_per_step": round(total_fl / 1e12, 4),
            "step_tflops_per_s": round(total_fl / 1e12 / (elapsed / a.steps), 2),
            "vae_decode_ms": round(vae_ms, 3) if vae_ms is not None else None,
            "vae_decode_tflop": round(vae_tflop, 3) if vae_tflop is not None else None,
            "sec_per_view": round((51 * ms_per_step + (vae_ms or 0.0)) / 1e3, 3),
            "sec_per_mesh_6_views_est": round(-(-6 // world) * (51 * ms_per_step + (vae_ms or 0.0)) / 1e3, 3),
            "roofline": {"bound": "mfma", "kernel": "k_gemm144 + k_gemm_pipe<...> + k_gemm8 (+ k_splitk_reduce): fp16 MFMA GEMM / implicit-GEMM conv3x3 of the UNet",
                         "achieved": round(achieved, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": gemm_n, "avg_launch_us": round(gemm_ms * 1e3 / max(gemm_n, 1), 2),
                         "flops_per_launch_avg": round(gemm_fl / max(gemm_n, 1) / 1e9, 3), "kernel_ms_per_step": round(gemm_ms, 3)},
            "attention": {"kernel": "k_attention_dma", "achieved": round(fl["attention"][1] / (att_ms * 1e-3) / 1e12, 2) if att_ms > 0 else 0.0,
                          "unit": "TFLOP/s", "launches_per_step": att_n, "kernel_ms_per_step": round(att_ms, 3)},
        }
        out["sec_per_mesh"] = round(mesh_s, 3) if mesh_s is not None else None
        out["sec_per_mesh_note"] = (f"{'MEASURED' if mesh_s is not None else 'NOT MEASURED (' + str(mesh_cover) + ')'}: one ConTEXTure.paint of {a.mesh_path}, {a.mesh_views} views over {world} rank(s), "
                                    + (f"full groups of {a.per_eval} views per lockstep evaluation (batch {2 * a.per_eval}), the rest " if a.per_eval > 1 else "")
                                    + f"{a.in_flight} views in flight per rank, 1200^2 render, 51 UNet evals + VAE decode per view, "
                                    + f"view weights, UV scatter, atlas merge; coverage {mesh_cover}")
        if two is not None:
            out["views_in_flight"] = two
            out["sec_per_mesh_6_views_est_2_in_flight"] = round(-(-6 // world) * (51 * two["2"]["ms_per_step_per_view"] + (vae_ms or 0.0)) / 1e3, 3)
        if atlas_ms is not None:
            out["atlas_allreduce_ms"] = round(atlas_ms, 3)
        if world == 1 and a.cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(S, forced=a.cpu_latent)
            except Exception as e:                      # never lose the GPU line to a host-side failure
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    # A rank whose mesh leg never came back sits in a collective (a GPU / RCCL hang): that is a FAILURE of the run, not a clean
    # one.  The steps/s line above is already out; every rank now learns through the rendezvous store (not through the process
    # group, whose queue the stuck collective blocks) whether ANY rank timed out, and if so all of them skip the teardown
    # (destroy_process_group against a dead peer has no deadline) and leave non-zero.
    if a.mesh and _any_rank_hung(dist, rank, world, mesh_hung):
        print(f"[bench] rank {rank}: mesh leg timed out on {'this rank' if mesh_hung else 'a peer'}; exiting 3", file=sys.stderr)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3)
    if dist is not None:
        dist.destroy_process_group()


def _any_rank_hung(dist, rank, world, mine, timeout_s=30.0):
    """Uniform decision across ranks over the TCP rendezvous store; a store failure (rank 0 already gone) counts as hung."""
    if dist is None or world == 1:
        return mine
    try:
        import datetime
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        store.set(f"ctx_mesh_leg_{rank}", "hung" if mine else "ok")
        keys = [f"ctx_mesh_leg_{r}" for r in range(world)]
        store.wait(keys, datetime.timedelta(seconds=timeout_s))
        return any(store.get(k) == b"hung" for k in keys)
    except Exception as e:
        print(f"[bench] rank {rank}: status exchange failed ({e})", file=sys.stderr)
        return True


if __name__ == "__main__":
    main()
