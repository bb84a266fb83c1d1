#!/usr/bin/env python3
"""bench.py - metaseg hot path on MI355X: DAPI images/s (1392x1040, 4-class metaseg).

One "step" = one pass of the whole device pipeline over a batch of synthetic 1040x1392 uint8 DAPI images that are
already resident in HBM: im2patches (35 tiles of 256x256 per image) -> U-Net (fp32, MFMA) -> stitch + uint8
quantise + argmax -> meta_inference clean-up -> connected-component ecDNA count, followed - when more than one
rank runs - by the path's only exchange: an all-gather of the per-image result records (RCCL over xGMI).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--images B] [--base 64]

``--gpus N`` with N > 1 starts the N ranks itself (one process per GPU, ``torch.distributed.run`` on 127.0.0.1) when
it is not already running under a launcher; under ``torch.distributed.run`` (RANK / WORLD_SIZE set) it is one rank.

Rank 0 prints ONE JSON line.  The workload is BASELINE.json configs[1] (single-GPU fp32 U-Net forward + argmax at
1392x1040) extended with the post-processing and count that the metric's "CCL ms/image" names; weights are the
canonical classic U-Net (base width 64, 23 conv layers, 96.2 GFLOP per 256x256 patch), random-initialised with a fixed
seed because metaseg.h5 is not distributable.

Order of work at N = 1: (1) the CPU-baseline worker processes are STARTED (idle) before this process touches the GPU - no
process is ever created after HIP is initialised; (2) the timed GPU region; (3) a host-inclusive leg (host arrays in,
labels + counts back on the host); (4) the CPU baseline legs on the waiting workers (after the GPU timing: 20 s of
all-core CPU load right before it cost 1.5 - 3 % of the GPU number); (5) full-size parity of the device results against
the CPU results of (4).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 1040, 1392
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak (no sparsity)
CPU_SEED0 = 900                    # synthetic image indices of the CPU-baseline / parity sample


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = CPU restatement of the reference path; the Keras original cannot run on this box)
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    """One image (or a patch sample of one image) through the oracle with a fixed number of torch threads."""
    base, idx, threads, n_patches, up = job
    import torch
    torch.set_num_threads(threads)
    from ecseg_amd import synth
    from oracle import pipeline as op
    from oracle import postproc, tiling, unet
    cfg = synth.unet_config(base=base, up=up)
    weights = synth.unet_weights(cfg, seed=0)
    im = synth.dapi_image(idx, H, W)
    pos = tiling.patch_positions(H, W)
    patches = tiling.extract_patches(im[..., None], pos)
    unet.forward(cfg, weights, patches[:1])                      # warm the thread pool / allocator (untimed)
    t0 = time.perf_counter()
    if n_patches:                                                # bounded sample: a few windows only
        preds = unet.forward(cfg, weights, patches[:n_patches])
        t1 = time.perf_counter()
        rng = np.random.default_rng(idx)
        full = np.concatenate([preds] * (-(-len(patches) // n_patches)))[:len(patches)]
        raw = op.raw_labels_from_probs(full[rng.permutation(len(patches))], pos)
    else:
        preds = np.concatenate([unet.forward(cfg, weights, patches[i:i + 7]) for i in range(0, len(patches), 7)])
        t1 = time.perf_counter()
        raw = op.raw_labels_from_probs(preds, pos)
    post = postproc.meta_inference(raw)
    nec = postproc.count_cc(post == 3)[0]
    t2 = time.perf_counter()
    if n_patches:
        return None, None, 0, t1 - t0, t2 - t1
    return raw.astype(np.uint8), post.astype(np.uint8), int(nec), t1 - t0, t2 - t1


def host_cpu_budget():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a 256-core box inside a
    16-CPU quota runs 16 busy processes at full speed and 64 at a quarter of it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q = open('/sys/fs/cgroup/cpu.max').read().split()                  # cgroup v2: "<quota|max> <period>"
        if q[0] != 'max':
            n = min(n, max(1, int(int(q[0]) / int(q[1]))))
    except (OSError, ValueError, IndexError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def _cpu_init(counter, threads):
    """Pool initializer: give every worker its own block of cores BEFORE torch creates its thread pool (64 workers x 4
    OpenMP threads otherwise end up bound to the same few cores and run 10x slower than one thread alone)."""
    with counter.get_lock():
        idx = counter.value
        counter.value += 1
    try:
        avail = sorted(os.sched_getaffinity(0))
        mine = avail[(idx * threads) % len(avail):][:threads] or avail[:threads]
        os.sched_setaffinity(0, set(mine))
    except (AttributeError, OSError):
        pass
    os.environ['OMP_NUM_THREADS'] = str(threads)
    os.environ['MKL_NUM_THREADS'] = str(threads)
    import torch
    torch.set_num_threads(threads)


def cpu_pools_start():
    """Start the CPU-baseline worker processes (idle until used).  Called BEFORE this process initialises HIP: no process
    is ever created from a process that has touched the GPU; the CPU legs themselves run after the timed GPU region so
    that 20 s of all-core CPU load do not precede it (measured: -1.5 ... -3 % on the GPU number otherwise)."""
    import multiprocessing as mp
    ncpu = host_cpu_budget()
    # one single-threaded worker per usable core, one image each: measured on the MI355X boxes of this pool
    # (tools/cpu_scale_probe.py) torch's CPU convolutions lose throughput with more threads per worker (16 x 8 threads:
    # 0.38 images/s, 16 x 1: 0.74) and the container's CPU quota ends the scaling near 16 busy workers (64 x 1: 0.52)
    threads = 1
    nproc = max(1, min(16, ncpu))                                 # bounded sample: at most 16 images
    ctx = mp.get_context('spawn')
    pool = ctx.Pool(nproc, initializer=_cpu_init, initargs=(ctx.Value('i', 0), threads))
    pool.map(_cpu_noop, range(nproc))                             # workers up, torch imported (untimed)
    return {'pool': pool, 'nproc': nproc, 'threads': threads, 'ncpu': ncpu}


def cpu_baseline(base, pools, up='transpose'):
    """(i) image-parallel workers over the host cores, one full image each; (ii) one thread on a bounded patch sample.
    Returns (cpu_baseline dict, single-thread dict, reference outputs for the parity check)."""
    pool, nproc, threads, ncpu = pools['pool'], pools['nproc'], pools['threads'], pools['ncpu']
    out = pool.map(_cpu_worker, [(base, CPU_SEED0 + i, threads, 0, up) for i in range(nproc)], chunksize=1)
    dt = max(o[3] + o[4] for o in out)                            # the workers run side by side; input synthesis is untimed
    refs = [(o[0], o[1], o[2]) for o in out]
    par = {'value': nproc / dt, 'unit': 'images/s', 'cores': nproc * threads, 'kind': 'port',
           'sample': '%d synthetic 1040x1392 images, one per worker process (%d processes x %d torch threads, each pinned to '
                     'its own cores; CPU budget of this container: %d), full path (U-Net via torch CPU fp32: %.1f s/image, stitch+argmax+'
                     'meta_inference+count via numpy/scipy: %.2f s/image), wall %.1f s'
                     % (nproc, nproc, threads, ncpu, float(np.mean([o[3] for o in out])), float(np.mean([o[4] for o in out])), dt)}
    n_sample = 2 if base >= 64 else 6 if base >= 32 else 18
    o = pool.map(_cpu_worker, [(base, CPU_SEED0, 1, n_sample, up)])[0]       # one worker busy, the others idle
    t_img = o[3] * 35.0 / n_sample + o[4]
    single = {'value': 1.0 / t_img, 'unit': 'images/s', 'cores': 1, 'kind': 'port',
              'sample': 'one thread: U-Net on %d of the 35 windows of one image (%.1f s, scaled x35/%d) + stitch/argmax/'
                        'meta_inference/count of one full image (%.2f s)' % (n_sample, o[3], n_sample, o[4])}
    pool.close()
    pool.join()
    return par, single, refs


def _cpu_noop(_):
    import torch                                                   # noqa: F401
    from oracle import pipeline, postproc, tiling, unet           # noqa: F401
    return 0


def parity_vs_cpu(hnd, refs):
    """Device path on the CPU sample's images.  Raw argmax labels may differ from the CPU's only where float rounding moves
    a quantised probability across a tie; everything after the raw labels is integer work and must be bit-exact."""
    from ecseg_amd import synth
    from oracle import postproc
    n = len(refs)
    imgs = np.stack([synth.dapi_image(CPU_SEED0 + i, H, W) for i in range(n)])
    g_raw, g_post, g_nec = hnd.segment_images(imgs, want_raw=True)
    raw_mis = [int((refs[i][0] != g_raw[i]).sum()) for i in range(n)]
    post_mis = [int((refs[i][1] != g_post[i]).sum()) for i in range(n)]
    k = min(n, 2)                                                  # the integer stages re-done on the CPU for two images
    exact = all(np.array_equal(postproc.meta_inference(g_raw[i].astype(np.int64)), g_post[i]) and
                int(postproc.count_cc(g_post[i] == 3)[0]) == int(g_nec[i]) for i in range(k))
    return {'sample': '%d synthetic 1040x1392 image(s)' % n, 'pixels': int(n * H * W),
            'raw_label_mismatch_px': int(sum(raw_mis)), 'post_label_mismatch_px': int(sum(post_mis)),
            'raw_label_mismatch_px_per_image': raw_mis,
            'n_ec_device': [int(v) for v in g_nec], 'n_ec_cpu': [int(r[2]) for r in refs],
            'integer_stages_bit_exact_on_device_raw_labels': bool(exact),
            'note': 'raw labels differ only where fp32 summation order moves a uint8-quantised probability across an '
                    'argmax tie (random-weight model = speckled, tie-rich output).  Both sides of this comparison are float32 '
                    'evaluations: against a float64 evaluation of the same network (profiles/r03_label_mismatch.json, 32 images) '
                    'this CPU oracle is wrong in 2.4 pixels per image and the device in 2.3, and where the two disagree float64 '
                    'sides with the device as often as with the oracle; on smooth-output / fitted models the counts are 0.4 vs 0.3 '
                    'and 0 vs 0.  Clean-up and counting are bit-exact functions of the raw labels'}


def aux_device_legs(hnd, local, n=64, with_comm=True):
    """the device cost of BASELINE configs[4] (`make meta_overlay`'s row: ecseg_overlay) and of
    meta_preprocess (ecseg_preprocess) on n resident synthetic FISH images - HIP events around the kernels alone inside the
    entry points (ecseg_get_timings()[ECSEG_T_COUNT]; host copies excluded) - with their algorithmic bytes and the fraction of
    the 8 TB/s HBM roof, plus ONE record all-gather through the C-ABI RCCL communicator with a single rank (the collective of
    the path has a latency figure even when the driver runs one GPU)."""
    import torch
    from ecseg_amd import synth
    from ecseg_amd._lib import Comm
    px = H * W
    base_rgb = [synth.dapi_image(700 + i, H, W, rgb=True) for i in range(8)]
    base_lab = [synth.label_map(700 + i, H, W) for i in range(8)]
    rgb = np.stack([np.roll(base_rgb[i % 8], (37 * (i // 8), 53 * (i // 8)), axis=(0, 1)) for i in range(n)])
    lab = np.stack([np.roll(base_lab[i % 8], (37 * (i // 8), 53 * (i // 8)), axis=(0, 1)) for i in range(n)])
    out = {}
    hnd.overlay(lab, rgb, 85)
    ms = []
    for _ in range(3):
        hnd.overlay(lab, rgb, 85)
        ms.append(hnd.timings()['count'])
    t = float(np.median(ms)) / n
    # 5 labellings x 9 B/px (image 1 + parents written 4 + read 4) + aux / mask passes (labels 1 + RGB 3 in, 1 out: 5 passes) +
    # 2 size filters (parents 4 + flags 2).  (Rounds 3 - 4a also counted 6 per-pixel flagged-root passes x 5 B/px = 162 MB in all; the
    # counts now come from the owner bits - 256 B per tile - and are not per-pixel work any more.)
    ov_bytes = px * (5 * 9 + 5 * 5 + 2 * 6)
    out['overlay_ms_per_image'] = {'value': round(t, 4), 'images': n, 'algorithmic_bytes_per_image': ov_bytes,
                                   'achieved_GBs': round(ov_bytes / (t * 1e-3) / 1e9, 1), 'frac_of_hbm_peak': round(ov_bytes / (t * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                                   'what': 'ecseg_overlay (src/meta_overlay.py:56-95): thresholds + 5 labellings + counts, realistic label maps (synth.label_map) + synthetic FISH RGB, kernels only'}
    hnd.preprocess(rgb)
    ms = []
    for _ in range(3):
        hnd.preprocess(rgb)
        ms.append(hnd.timings()['count'])
    t = float(np.median(ms)) / n
    pp_bytes = px * (3 + 1 + 1)           # RGB in, gray out, gray re-read by the histogram (+2 for an inverted image: none here)
    out['preprocess_ms_per_image'] = {'value': round(t, 4), 'images': n, 'algorithmic_bytes_per_image': pp_bytes,
                                      'achieved_GBs': round(pp_bytes / (t * 1e-3) / 1e9, 1), 'frac_of_hbm_peak': round(pp_bytes / (t * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                                      'what': 'ecseg_preprocess (src/image_tools.py:86-96): blue channel + 256-bin histogram + Otsu + invert, uint8 RGB, kernels only'}
    # "CCL ms/image" on a REALISTIC label map beside the headline's speckle figure (random weights give 16 - 17 k
    # components per image): the clean-up + count of ecseg_meta_inference_dev on seeded synth.label_map images (blobs + 0.2 % salt),
    # independent of any network, kernels only
    d_in = torch.from_numpy(np.ascontiguousarray(lab)).to('cuda:%d' % local)
    d_out = torch.empty_like(d_in)
    nec = torch.zeros(n, dtype=torch.int32, device=d_in.device)
    hnd.meta_inference_dev(d_in.data_ptr(), n, H, W, d_out.data_ptr(), nec.data_ptr())
    ms = []
    for _ in range(5):
        hnd.meta_inference_dev(d_in.data_ptr(), n, H, W, d_out.data_ptr(), nec.data_ptr())
        ms.append(hnd.timings()['post'])
    t = float(np.median(ms)) / n
    cc_bytes = px * (7 * 9 + 7 * 2)                     # 7 labellings x 9 B/px + 7 stencil passes x 2 B/px (111 MB per 1040 x 1392 image)
    out['ccl_ms_per_image_realistic'] = {'value': round(t, 4), 'images': n, 'algorithmic_bytes_per_image': int(cc_bytes),
                                         'achieved_GBs': round(cc_bytes / (t * 1e-3) / 1e9, 1), 'frac_of_hbm_peak': round(cc_bytes / (t * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                                         'mean_n_ec': float(nec.float().mean().item()),
                                         'what': 'ecseg_meta_inference_dev (src/image_tools.py:15-84 + src/metaseg.py:46) on synth.label_map images (blobs + 0.2 % salt noise), kernels only; '
                                                 'the headline ccl_ms_per_image is the same kernels on the speckled argmax of the random-weight bench model'}
    del d_in, d_out, nec
    if not with_comm:                                              # (tools/aux_legs.py under rocprofv3: RCCL's own profiler hooks crash there)
        return out
    try:
        comm = Comm(Comm.unique_id(), 0, 1, local)
        rec = torch.zeros((n, 16), dtype=torch.int64, device='cuda:%d' % local)
        got = torch.empty_like(rec)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            comm.allgather_records_dev(rec.data_ptr(), n, got.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            comm.allgather_records_dev(rec.data_ptr(), n, got.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        out['allgather_one_rank_us'] = {'value': round((time.perf_counter() - t0) / reps * 1e6, 1), 'records': n,
                                        'what': 'ecseg_allgather_records_dev (ncclAllGather of %d x 128-B records) on a 1-rank communicator, enqueue + completion, mean of %d' % (n, reps)}
        comm.close()
    except Exception as e:                                         # RCCL not loadable on this box
        out['allgather_one_rank_us'] = {'value': None, 'error': str(e)}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """--gpus N > 1 outside a launcher: start one rank per GPU and relay rank 0's JSON line.  This parent never touches
    the GPU (device_count() does not initialise HIP)."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit('bench.py: --gpus %d but only %d HIP device(s) visible' % (args.gpus, have))
    # --standalone: torch.distributed.run hosts its own c10d rendezvous store on a port the store itself binds (no
    # bind-then-close race with other jobs on the node); --local-addr pins it to 127.0.0.1 (the hostname may not resolve)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL's intra-node transport shares device buffers between the rank processes through HIP
    # IPC handles; this pool's host driver supports only the dmabuf flavour, and with the legacy mode left on
    # hipIpcGetMemHandle fails with "invalid argument" as soon as a second rank joins
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    sys.exit(subprocess.run(cmd, env=env).returncode)


KIND_NAMES = ('conv_mfma_kernel', 'conv_wino_kernel F(2x2)', 'conv_wino4r_kernel / conv_wino4_kernel F(4x4)',
              'conv_wino_res_kernel F(2x2), filter-resident', 'conv_wino16_kernel F(2x2), 16x16x4 MFMA',
              'conv_wino4s_kernel F(4x4), 3-way bf16 split operands on the bf16 pipe', 'convs_kernel one-tap GEMM, 3-way bf16 split operands on the bf16 pipe')
KIND_ISSUED = (1.0, 16.0 / 36.0, 0.25, 16.0 / 36.0, 16.0 / 36.0, 0.25, 1.0)      # multiplies issued / multiplies of the direct 3x3 convolution (kind 5: each as 6 bf16 products)


def aggregate(recs):
    agg = {}
    for r in recs:
        a = agg.setdefault(r['op'], dict(kind=r['kind'] & 0xff, pool=bool(r['kind'] & 0x100), head=bool(r['kind'] & 0x200),
                                         first=bool(r['kind'] & 0x400), launches=0, ms=0.0, flops=0.0, executed=0.0))
        a['launches'] += 1; a['ms'] += r['ms']; a['flops'] += r['flops']; a['executed'] += r['executed_flops']
    return agg


def layer_table(model, recs, steps):
    """Per-convolution table from the per-launch HIP-event records of the timed steps."""
    plan = model.plan
    agg = aggregate(recs)
    names = {v: k for k, v in plan.layer_tensor.items()}
    rows = []
    for op, a in sorted(agg.items()):
        o = plan.ops[op]
        ti, to = plan.tensors[o['in0']], plan.tensors[o['out']]
        ms = a['ms'] / a['launches']
        rows.append({'op': op, 'layer': names.get(o['out'], '?'), 'type': 'conv%dx%d' % (o['kh'], o['kw']) if o['op'] == 1 else 'convT%dx%d' % (o['kh'], o['kw']),
                     'in': [ti['h'], ti['w'], ti['c']], 'out': [to['h'], to['w'], to['c']],
                     'kernel': KIND_NAMES[a['kind']] + (' + fused 2x2 max-pool' if a['pool'] else '') + (' + fused 1x1 head' if a['head'] else '') +
                               (' + first 3x3 convolution computed into the halo' if a['first'] else ''),
                     'launches': a['launches'], 'avg_ms': round(ms, 4),
                     'algorithmic_gflop_per_launch': round(a['flops'] / a['launches'] / 1e9, 2),
                     'executed_gflop_per_launch': round(a['executed'] / a['launches'] / 1e9, 2),
                     'algorithmic_tflops': round(a['flops'] / a['ms'] / 1e9, 1),
                     'executed_tflops': round(a['executed'] / a['ms'] / 1e9, 1),
                     'executed_frac_of_peak': round(a['executed'] / a['ms'] / 1e9 / PEAK_FP32_MFMA_TFLOPS, 3)})
    return rows


def roofline_8d(plan, recs, steps, patches_per_step, unet_ms_per_step):
    """SURVEY 8d's mixed roofline of the whole U-Net: sum over its convolution layers of max(F_l / 157.3e12, B_l / 8.0e12)
    divided by the measured U-Net time.  F_l = FLOPs the kernel EXECUTES for the layer (Winograd F(4x4) issues 36/144, F(2x2)
    16/36 of a 3x3 convolution's multiplies; cropped launches count the regions computed; layers on VALU kernels - first
    convolution, unfused head - count their direct FLOPs and are HBM-bound anyway); B_l = algorithmic fp32 bytes, every
    tensor read / written once (4 N (Hin Win Cin + Hout Wout Cout) + the kernel), scaled by the computed fraction of a
    cropped launch; a 1x1 head finished by the previous convolution's output stage counts neither its own bytes nor the
    write of the tensor it would have read, and a first convolution computed into the next layer's halo (record bit 0x400:
    its FLOPs are in that launch's) leaves only its one-channel input and its filter - the tensor between the two never
    exists.  frac <= 1: the ideal schedule of the same layers on the same algorithms."""
    agg = aggregate(recs)
    fused_heads = {op + 1 for op, a in agg.items() if a['head']}
    fused_firsts = {op - 1 for op, a in agg.items() if a['first']}
    ideal_f = ideal_b = tot_f = tot_b = 0.0
    for k, o in enumerate(plan.ops):
        if o['op'] not in (1, 2) or k in fused_heads:
            continue
        ti, to = plan.tensors[o['in0']], plan.tensors[o['out']]
        n = patches_per_step
        px = to['h'] * to['w'] if o['op'] == 1 else ti['h'] * ti['w']
        F = 2.0 * o['kh'] * o['kw'] * ti['c'] * to['c'] * px * n
        b_in, b_out = 4.0 * n * ti['h'] * ti['w'] * ti['c'], 4.0 * n * to['h'] * to['w'] * to['c']
        b_w = 4.0 * plan.weights[o['w0']].size
        if k in fused_firsts:                       # executed inside op k + 1's launch: its input and filter are all that moves
            tot_b += b_in + b_w; ideal_b += (b_in + b_w) / (PEAK_HBM_GBS * 1e9)
            continue
        a = agg.get(k)
        if a is not None:
            issued = a['flops'] * KIND_ISSUED[a['kind']]
            computed = min(1.0, a['executed'] / issued) if issued else 1.0
            F = a['executed'] / steps
            if a['head']:
                b_out = 0.0
            if a['first']:
                b_in = 0.0
            b_in, b_out = b_in * computed, b_out * computed
        B = b_in + b_out + b_w
        tf, tb = F / (PEAK_FP32_MFMA_TFLOPS * 1e12), B / (PEAK_HBM_GBS * 1e9)
        tot_f += F; tot_b += B
        if tf >= tb:
            ideal_f += tf
        else:
            ideal_b += tb
    ideal = ideal_f + ideal_b
    return {'frac': round(ideal * 1e3 / unet_ms_per_step, 4), 'ideal_unet_ms_per_step': round(ideal * 1e3, 4),
            'measured_unet_ms_per_step': round(unet_ms_per_step, 4),
            'bound': 'mfma' if ideal_f >= ideal_b else 'hbm',
            'ideal_ms_in_mfma_bound_layers': round(ideal_f * 1e3, 4), 'ideal_ms_in_hbm_bound_layers': round(ideal_b * 1e3, 4),
            'executed_tflop_per_step': round(tot_f / 1e12, 4), 'algorithmic_gbyte_per_step': round(tot_b / 1e9, 3),
            'peaks': {'fp32_mfma_tflops': PEAK_FP32_MFMA_TFLOPS, 'hbm_gbs': PEAK_HBM_GBS},
            'definition': 'sum_l max(F_l / peak_mfma, B_l / peak_hbm) / t_unet; F_l executed FLOPs, B_l algorithmic bytes (SURVEY 8d)'}


class DeviceRun:
    """One synthetic model on this rank's GPU: inputs resident in HBM, `step()` = the whole device pipeline over B images."""

    def __init__(self, base, B, group, local, rank, world, args, up=None, batchnorm=False):
        import torch
        from ecseg_amd import dist as edist
        from ecseg_amd import synth
        from ecseg_amd.model import MetasegModel
        self.torch, self.edist = torch, edist
        self.base, self.B, self.world = base, B, world
        dev = torch.device('cuda', local)
        cfg = synth.unet_config(base=base, up=up or args.up, batchnorm=batchnorm)
        self.model = MetasegModel(cfg, synth.unet_weights(cfg, seed=0), device=local)
        hnd = self.hnd = self.model.handle
        hnd.set_images_per_group(group)
        hnd.set_option('overlap_post', 1 if args.overlap else 0)
        if args.wino is not None:
            hnd.set_option('winograd', args.wino)
        for kv in args.opt:
            k, v = kv.split('=')
            hnd.set_option(k, int(v))
        total = B * world                              # weak scaling: per-GPU work fixed
        self.total_images = total
        start, stop, per = edist.shard_bounds(total, rank, world)
        self.host = np.stack([synth.dapi_image(i, H, W) for i in range(start, stop)])
        self.gray = torch.from_numpy(self.host).to(dev)
        self.raw = torch.empty_like(self.gray)
        self.post = torch.empty_like(self.gray)
        self.nec = torch.zeros(B, dtype=torch.int32, device=dev)
        self.rec = torch.from_numpy(edist.make_records(start, stop - start, per)).to(dev)
        self.comm, self.gathered, self.allgather_via = None, None, 'none (1 rank)'
        self.ag_events = []
        if world > 1 or (os.environ.get('ECSEG_BENCH_FORCE_COMM') and torch.distributed.is_initialized()):   # (the latter: 1-GPU test of the same code)
            self._setup_c_abi_collective(local, rank, world)

    def _setup_c_abi_collective(self, local, rank, world):
        """The record all-gather through the C ABI (ecseg_allgather_records_dev: RCCL resolved by the library itself) instead
        of torch.distributed; the unique id travels over the torch process group that the launcher contract provides anyway.
        Verified once against torch's all_gather; every rank must agree, otherwise all fall back to torch (and say so)."""
        import torch.distributed as dist
        from ecseg_amd._lib import Comm
        torch = self.torch
        ok, why = 1, ''
        box = [None]
        if rank == 0:
            try:
                box[0] = Comm.unique_id()
            except Exception as e:                                           # librccl not loadable, ...
                why = str(e)
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            ok = 0
        else:
            try:
                self.comm = Comm(box[0], rank, world, local)
                self.gathered = torch.empty((world * self.rec.shape[0], self.rec.shape[1]), dtype=torch.int64, device=self.rec.device)
                torch.cuda.synchronize()
                self.comm.allgather_records_dev(self.rec.data_ptr(), self.rec.shape[0], self.gathered.data_ptr())
                want = self.edist.allgather_records(self.rec)
                ok = int(bool(torch.equal(self.gathered, want)))
                why = '' if ok else 'result differs from torch.distributed.all_gather_into_tensor'
            except Exception as e:
                ok, why = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=self.rec.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            self.allgather_via = 'ecseg_allgather_records_dev (C ABI; RCCL loaded by libecseg_hip.so), checked against torch.distributed once'
        else:
            self.comm = None
            self.allgather_via = 'torch.distributed.all_gather_into_tensor (C-ABI collective unavailable: %s)' % (why or 'another rank failed')

    def step(self):
        self.hnd.segment_images_dev(self.gray.data_ptr(), self.B, H, W, self.raw.data_ptr(), self.post.data_ptr(), self.nec.data_ptr())
        self.rec[:self.B, self.edist.F_NEC] = self.nec.to(self.torch.int64)
        timed_ag = self.world > 1 or self.comm is not None   # the all-gather as its own stage (events on torch's current stream)
        if timed_ag:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.comm is not None:
            self.comm.allgather_records_dev(self.rec.data_ptr(), self.rec.shape[0], self.gathered.data_ptr(),
                                            stream=self.torch.cuda.current_stream().cuda_stream)
            out = self.gathered
        else:
            out = self.edist.allgather_records(self.rec)
        if timed_ag:
            e1.record()
            self.ag_events.append((e0, e1))
        return out

    def barrier(self):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup, profile=True):
        """W untimed + K timed steps bracketed by barrier + synchronize; MAX over ranks.  -> dict of raw measurements."""
        import torch.distributed as dist
        hnd = self.hnd
        for _ in range(warmup):
            self.step()
        hnd.set_kernel_profiling(profile)
        self.ag_events = []
        stage = {k: 0.0 for k in hnd.T_NAMES}
        conv_ms = conv_flops = conv_exec = 0.0
        conv_launches = 0
        recs = []
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = self.step()
            for k, v in hnd.timings().items():
                stage[k] += v
            ms, nl, fl = hnd.conv_profile()
            conv_ms += ms; conv_launches += nl; conv_flops += fl
            conv_exec += hnd.conv_executed_flops()
            if profile:
                recs += hnd.conv_launch_profile()
        self.barrier()
        dt = time.perf_counter() - t0
        hnd.set_kernel_profiling(False)
        tmax = self.torch.tensor([dt], dtype=self.torch.float64, device=self.gray.device)
        if dist.is_initialized():
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if self.ag_events:                                    # (after the closing barrier + synchronize: every event has completed)
            stage = dict(stage, allgather=sum(a.elapsed_time(b) for a, b in self.ag_events))
        return dict(dt=float(tmax.item()), stage=stage, conv_ms=conv_ms, conv_launches=conv_launches, conv_flops=conv_flops,
                    conv_exec=conv_exec, recs=recs, out=out, steps=steps)

    def single_image_latency(self, reps=10):
        """BASELINE configs[1] read literally: ONE 1040x1392 image resident in HBM -> labels + count, synchronous."""
        call = lambda: self.hnd.segment_images_dev(self.gray.data_ptr(), 1, H, W, self.raw.data_ptr(), self.post.data_ptr(), self.nec.data_ptr())
        call(); call()
        self.torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            call()
            self.torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return {'median_ms': round(float(np.median(ts)), 3), 'min_ms': round(min(ts), 3), 'reps': reps,
                'what': 'ecseg_segment_images_dev on ONE resident 1040x1392 image (35 windows): tile -> U-Net -> stitch/argmax -> '
                        'meta_inference -> count, host-synchronous'}

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.hnd.close()
        del self.gray, self.raw, self.post, self.nec, self.rec
        self.torch.cuda.empty_cache()


def split_leg(run, steps, headline, wino_mode):
    """the same step with the F(4x4) channel sums on the bf16 matrix pipe and both operands split exactly into three
    bf16 pieces (option winograd = 3, conv_wino4s_kernel; float32-accurate, float32 accumulate).  Reported beside - never instead of -
    the fp32-MFMA headline: images/s, the split kernel's own rate against the BF16 peak (each fp32-equivalent product is issued as
    six bf16 products), max |dp| and raw-label differences against the fp32-MFMA path on two images, and - base 64 with the committed
    float64 adjudicator fixture - how often each path is wrong on the fixture's hard pixels."""
    hnd = run.hnd                         # (a handle of its own: timed on the headline's handle right after switching modes the same step read 4 % slower
                                          # in wall time than its own stage timers said - 135.7 vs 124.8 ms - and than a fresh `--wino 3` run measured)
    out = {'what': 'option winograd = 3: conv_wino4s_kernel on every 3x3 layer with whole 64-channel output blocks and convs_kernel on the 2x2 / stride-2 up-convolutions '
                   '(v_mfma_f32_32x32x16_bf16, 3-way exact bf16 split of both operands, 6 of the 9 piece products, float32 accumulate); everything else as in the headline run'}
    try:
        hnd.set_option('winograd', 3)
        m = run.timed(steps, 1, profile=True)
        s = model_summary(run, m)
        r5 = [r for r in m['recs'] if (r['kind'] & 0xff) in (5, 6)]
        ms5, ex5 = sum(r['ms'] for r in r5), sum(r['executed_flops'] for r in r5)
        out.update({'value': s['value'], 'unit': 'images/s', 'ms_per_step': s['ms_per_step'], 'vs_f32_headline': round(s['value'] / headline, 4),
                    'stage_ms_per_image': s['stage_ms_per_image'], 'dtype': 'bf16x3 products, f32 accumulate (f32-accurate)'})
        if ms5 > 0:
            eq = ex5 / (ms5 * 1e-3) / 1e12
            out['split_kernel'] = {'launches_per_step': len(r5) // steps, 'ms_per_step': round(ms5 / steps, 3),
                                   'fp32_equivalent_executed_tflops': round(eq, 2), 'bf16_product_tflops': round(6 * eq, 2),
                                   'frac_of_bf16_mfma_peak': round(6 * eq / PEAK_BF16_MFMA_TFLOPS, 4), 'peak': PEAK_BF16_MFMA_TFLOPS,
                                   'vs_fp32_mfma_peak': round(eq / PEAK_FP32_MFMA_TFLOPS, 4),
                                   'note': 'executed = Winograd F(4x4) products really issued (36/144 of the direct convolution, cropped regions only); '
                                           'the kernel is bound by the LDS-DMA path that streams its filter (110 KB per 8-channel group and workgroup, '
                                           'tools/micro/ldsdma_rate.hip), not by the matrix pipe: DESIGN.md 5.1b'}
        pair = run.host[:2]
        a = hnd.segment_images(pair, want_raw=True, want_probs=True)
        hnd.set_option('winograd', 2)
        b = hnd.segment_images(pair, want_raw=True, want_probs=True)
        out['vs_fp32_mfma_path'] = {'images': 2, 'max_abs_dp': float(np.abs(a[3] - b[3]).max()), 'raw_label_px_different': int((a[0] != b[0]).sum()),
                                    'n_ec_different': int((np.asarray(a[2]) != np.asarray(b[2])).sum())}
        fx = os.path.join(ROOT, 'tests', 'golden', 'label_truth_random_base64.npz')
        if run.base == 64 and os.path.exists(fx):
            z = np.load(fx)
            from ecseg_amd import synth
            n = 8
            imgs = np.stack([synth.dapi_image(int(z['seed0']) + i) for i in range(n)])
            adj = {'images': n, 'hard_px': int(sum(len(z['idx_%d' % i]) for i in range(n))),
                   'float32_cpu_oracle_wrong': int(z['oracle32_wrong_on_hard_px'][:n].sum()),
                   'fixture': 'tests/golden/label_truth_random_base64.npz (float64 labels of the pixels within 2.55e-5 of a change of the quantised argmax)'}
            for mode, name in ((3, 'split_bf16x3_wrong'), (2, 'fp32_mfma_wrong')):
                hnd.set_option('winograd', mode)
                raw = hnd.segment_images(imgs, want_raw=True)[0]
                adj[name] = int(sum((raw[i].ravel()[z['idx_%d' % i].astype(np.int64)] != z['truth_%d' % i]).sum() for i in range(n)))
            out['float64_adjudicator'] = adj
    finally:
        hnd.set_option('winograd', wino_mode)
    return out


def model_summary(run, m):
    """The per-model part of the JSON line (used for the headline model and for the narrow models)."""
    steps, B = m['steps'], run.B
    unet_ms_step = m['stage']['unet'] / steps
    res = {'value': round(run.total_images * steps / m['dt'], 3), 'unit': 'images/s', 'images_per_gpu_per_step': B, 'steps': steps,
           'ms_per_step': round(m['dt'] / steps * 1e3, 3), 'unet_base': run.base,
           'gflop_per_patch': round(run.model.plan.flops_per_patch() / 1e9, 2),
           'stage_ms_per_image': {k: round(v / (steps * B), 4) for k, v in m['stage'].items()},
           'ccl_ms_per_image': round(m['stage']['post'] / (steps * B), 4)}
    if m['conv_launches'] and m['recs']:
        exe = m['conv_exec'] / (m['conv_ms'] * 1e-3) / 1e12
        res['mfma_executed_tflops'] = round(exe, 2)
        res['mfma_executed_frac_of_peak'] = round(exe / PEAK_FP32_MFMA_TFLOPS, 4)
        res['roofline_8d'] = roofline_8d(run.model.plan, m['recs'], steps, B * 35, unet_ms_step)
        res['roofline'] = {'bound': res['roofline_8d']['bound'], 'achieved': round(exe, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': round(exe / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': None,
                           'kernel': 'all MFMA-convolution launches of the step (HIP events on the handle\'s stream), executed FLOPs'}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--images', type=int, default=16, help='images per GPU per step')
    ap.add_argument('--base', type=int, default=64, help='U-Net base width of the synthetic model')
    ap.add_argument('--up', default='transpose', help="decoder up-sampler of the synthetic U-Net: transpose (2x2 / stride 2, the canonical model), transpose3 / transpose4 (3x3 / 4x4 at stride 2), upsample")
    ap.add_argument('--group', type=int, default=16, help='images per internal U-Net launch group (0: automatic)')
    ap.add_argument('--overlap', action='store_true', help='run post-processing on a second stream')
    ap.add_argument('--direct', action='store_true', help='direct implicit-GEMM 3x3 kernel instead of Winograd')
    ap.add_argument('--wino', type=int, default=None, help='3x3 kernel: 0 direct, 1 Winograd F(2x2,3x3), 2 F(4x4,3x3) where eligible (default: library default)')
    ap.add_argument('--opt', action='append', default=[], help='library tuning knob key=value (ecseg_set_option), repeatable')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-host-inclusive', action='store_true')
    ap.add_argument('--no-narrow', action='store_true', help='skip the base-32 / base-16 legs and the single-image latency (N = 1 only)')
    ap.add_argument('--layer-table', default=None, help='write the per-layer roofline table (JSON) to this path')
    args = ap.parse_args()

    under_launcher = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if args.gpus > 1 and not under_launcher:
        self_launch(args)
    want_cpu = args.gpus <= 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1 and not args.no_cpu_baseline
    pools = cpu_pools_start() if want_cpu else None              # worker processes exist before this process initialises HIP

    import torch                      # first: libecseg_hip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist
    from ecseg_amd import dist as edist

    rank, world = edist.init_process_group()
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != max(1, args.gpus):
        raise SystemExit('bench.py: --gpus %d but the launcher started %d rank(s)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no HIP device visible); there is no CPU fallback')
    torch.cuda.set_device(local)
    if args.direct:
        args.wino = 0
    wino_mode = 2 if args.wino is None else max(0, min(3, args.wino))
    B = args.images

    run = DeviceRun(args.base, B, args.group, local, rank, world, args)
    m = run.timed(args.steps, args.warmup, profile=not args.no_kernel_profile)
    hnd, model = run.hnd, run.model

    if rank == 0:
        counts = edist.compact_records(m['out'])[:, edist.F_NEC]
        assert len(counts) == run.total_images
        summ = model_summary(run, m)
        res = {
            'metric': 'DAPI images/sec (1392x1040, 4-class metaseg)', 'value': summ['value'], 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': summ['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1] + post-process: %d synthetic 1040x1392 uint8 DAPI images per GPU per '
                                   'step, 35 tiles of 256x256 each, canonical U-Net base %d%s (%.1f GFLOP/patch, seeded random '
                                   'weights) -> stitch/uint8-quantise/argmax -> meta_inference -> ecDNA count'
                                   % (B, args.base, '' if args.up == 'transpose' else ' with %s up-convolutions' % args.up, summ['gflop_per_patch']),
                       'images_per_gpu_per_step': B, 'unet_base': args.base, 'patches_per_image': 35,
                       'parallelism': 'image-parallel x%d, all-gather of 128-B records' % world, 'allgather': run.allgather_via},
            'stage_ms_per_image': summ['stage_ms_per_image'], 'ccl_ms_per_image': summ['ccl_ms_per_image'],
        }
        if m['conv_launches']:
            conv_ms, conv_launches, conv_flops, conv_exec = m['conv_ms'], m['conv_launches'], m['conv_flops'], m['conv_exec']
            # HBM bytes per launch from a committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary (tools/profile_round.sh) - only
            # one of THIS configuration (base, up-sampler, 3x3 kernel) measured on THESE kernel sources; anything else is stale
            traffic = traffic_src = None
            traffic_stale = False
            try:
                from ecseg_amd.build import source_hash
                want = {'base': str(args.base), 'up': args.up, 'wino': str(wino_mode)}
                cur = source_hash()
                for f in sorted((f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_pmc_traffic.json')), reverse=True):
                    d = json.load(open(os.path.join(ROOT, 'profiles', f)))
                    if d.get('config') != want:
                        continue
                    if d.get('kernel_source_sha256') != cur:
                        traffic_stale = True
                        continue
                    traffic = d['conv_mfma_all']['hbm_bytes_per_launch']
                    traffic_src = 'profiles/' + f + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these kernel sources, read from the committed summary, not measured in this run)'
                    traffic_stale = False
                    break
            except Exception:
                traffic = None
            up_desc = {'transpose': '2x2 up-convs', 'transpose3': '3x3 / stride-2 up-convs, sub-pixel form', 'transpose4': '4x4 / stride-2 up-convs, sub-pixel form',
                       'upsample': 'no up-convs: UpSampling2D'}.get(args.up, args.up)
            alg = conv_flops / (conv_ms * 1e-3) / 1e12
            exe = conv_exec / (conv_ms * 1e-3) / 1e12
            r8 = summ.get('roofline_8d')
            res['roofline'] = {'bound': r8['bound'] if r8 else 'mfma',
                               'kernel': {0: 'conv_mfma_kernel (direct implicit GEMM)',
                                          1: 'conv_wino_kernel (Winograd F(2x2,3x3)) + conv_mfma_kernel (%s)' % up_desc,
                                          2: 'conv_wino4r_kernel (Winograd F(4x4,3x3), cooperative row pass; conv_wino4_kernel on the 32-channel layer) + conv_mfma_kernel (%s)' % up_desc,
                                          3: 'conv_wino4s_kernel (Winograd F(4x4,3x3), bf16x3 split operands) + conv_mfma_kernel (%s)' % up_desc}[wino_mode] +
                                         ', fp32 v_mfma_f32_32x32x2_f32',
                               'achieved': round(exe, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(exe / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_stale': traffic_stale, 'traffic_source': traffic_src,
                               'avg_launch_ms': round(conv_ms / conv_launches, 4), 'launches': int(conv_launches),
                               'executed_flop_per_launch_avg': conv_exec / conv_launches,
                               'algorithmic_flop_per_launch_avg': conv_flops / conv_launches,
                               'algorithmic_tflops': round(alg, 2),
                               'algorithmic_over_direct_peak': round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                               'note': 'achieved / frac = FLOPs the matrix cores EXECUTED (multiplies really issued: Winograd '
                                       'F(4x4,3x3) issues 36/144, F(2x2) 16/36 of a 3x3 convolution\'s multiplies; cropped decoder '
                                       'layers count only the 16x16 regions computed) / HIP-event time of every MFMA-conv launch '
                                       'on the handle\'s stream / fp32 MFMA peak, so frac <= 1 is matrix-pipe utilisation.  '
                                       'algorithmic_tflops = direct-convolution FLOPs of whole 256x256 windows (SURVEY 8d) / the '
                                       'same time; it exceeds the direct-convolution roof by the Winograd and cropping factors.  '
                                       'bound = the term that dominates roofline_8d for this model'}
            if r8:
                res['roofline_8d'] = r8
            if args.layer_table and m['recs']:
                rows = layer_table(model, m['recs'], args.steps)
                json.dump({'command': ' '.join(sys.argv), 'patches_per_launch': B * 35, 'peak_tflops': PEAK_FP32_MFMA_TFLOPS,
                           'layers': rows}, open(args.layer_table, 'w'), indent=1)
        if world == 1 and not args.no_host_inclusive:
            # host arrays in (pageable numpy), labels + counts back on the host: H2D + device pipeline + D2H per call
            n_hi = 2
            hnd.segment_images(run.host, want_raw=False)
            t1 = time.perf_counter()
            for _ in range(n_hi):
                hnd.segment_images(run.host, want_raw=False)
            hi = (time.perf_counter() - t1) / n_hi
            res['host_inclusive'] = {'value': round(B / hi, 3), 'unit': 'images/s', 'ms_per_image': round(hi / B * 1e3, 3),
                                     'what': 'ecseg_segment_images: %d uint8 images from pageable host memory (H2D), device '
                                             'pipeline, post-processed labels + counts back to host memory (D2H), synchronous'
                                             % B}
        if world == 1 and not args.no_narrow and wino_mode == 2:
            r4 = DeviceRun(args.base, B, args.group, local, rank, world, args)
            res['split_bf16x3'] = split_leg(r4, args.steps, summ['value'], wino_mode)
            r4.close()
        if world == 1 and not args.no_narrow:
            res.update(aux_device_legs(hnd, local))
            res['single_image_latency_ms'] = run.single_image_latency()
            # "also run base 32 and 16" - the same pipeline on the narrower canonical models with the automatic
            # launch-group size (as many images per U-Net launch as fit ~48 GB of activations: 32 / 64 images), one step =
            # that many images; before the CPU legs (all-core CPU load right before a GPU timing costs it 1.5 - 3 %)
            narrow = {}
            for nb, nimg in ((32, 32), (16, 64)):
                if nb == args.base:
                    continue
                r2 = DeviceRun(nb, nimg, 0, local, rank, world, args)
                narrow['base%d' % nb] = model_summary(r2, r2.timed(5, 1, profile=True))
                r2.close()
            res['narrow_models'] = narrow
            # the decoder family of the most common public Keras U-Net - UpSampling2D(2) + Conv2D(2x2, 'same') instead of
            # Conv2DTranspose - lowered (round 6) to ONE 3x3 / stride-2 transposed convolution with pre-summed taps on the un-upsampled tensor
            ups = {}
            for tag, nb, nimg, bn in (('upsample_base64', 64, 16, False), ('upsample_base16', 16, 64, False), ('upsample_base16_batchnorm', 16, 64, True)):
                r3 = DeviceRun(nb, nimg, 0, local, rank, world, args, up='upsample', batchnorm=bn)
                ups[tag] = model_summary(r3, r3.timed(5, 1, profile=True))
                r3.close()
            res['upsample_decoder_models'] = ups
        if pools is not None:
            cpu_par, cpu_single, refs = cpu_baseline(args.base, pools, args.up)    # after the timed region, on the idle workers
            res['cpu_baseline'] = cpu_par
            res['cpu_baseline_single_thread'] = cpu_single
            res['parity_vs_cpu'] = parity_vs_cpu(hnd, refs)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
