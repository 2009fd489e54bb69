#!/usr/bin/env python3
"""HBM-SIDE traffic of the GEMM launches, to settle what the fabric-side counters cannot.

rocprofv3's FETCH_SIZE / WRITE_SIZE come from the L2's memory-side request counters (TCC_EA0_RDREQ / WRREQ): requests
that leave an XCD's L2 -- Infinity-Cache (MALL) hits included (MI355X_MICROARCH.md, HBM section).  The memory
controllers' own activity is what `/sys/class/drm/card*/device/mem_busy_percent` (the SMU's UMC busy figure) reports.
This tool samples it while ONE kernel runs in a loop and converts it to bytes per launch with a calibration taken in
the same process:

  * calibration A: `npm_add` over three 1 GiB tensors (12 B per element from / to HBM, nothing cacheable): the busy
    figure per TB/s of real HBM traffic;
  * calibration B: the same kernel over three 16 MiB tensors that live in the 256 MiB Infinity Cache: its L2-miss
    traffic is the same per element, its HBM traffic is ~0 -- if the busy figure stays near 0 there, the figure is
    HBM-side and excludes Infinity-Cache hits (which is the property this measurement needs);
  * then each GEMM shape of interest (`--shapes`), >= `--seconds` each.

    python tools/hbm_side.py [--seconds 3] > profiles/rNN_hbm_side.log
"""

import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def busy_file():
    for path in sorted(glob.glob('/sys/class/drm/card*/device/mem_busy_percent')):
        try:
            with open(path) as f:
                int(f.read())
            return path
        except (OSError, ValueError):
            continue
    return None


class Sampler(threading.Thread):
    def __init__(self, path, period=0.01):
        super().__init__(daemon=True)
        self.path, self.period, self.values, self.stop_flag = path, period, [], False

    def run(self):
        while not self.stop_flag:
            try:
                with open(self.path) as f:
                    self.values.append(int(f.read()))
            except (OSError, ValueError):
                pass
            time.sleep(self.period)


def measure(path, fn, D, seconds):
    """(mean busy % over the steady part, launches per second)."""
    fn()
    D.synchronize()
    t0 = time.perf_counter()
    fn()
    D.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    batch = max(1, int(0.05 / one))
    sampler = Sampler(path)
    sampler.start()
    launches, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(batch):
            fn()
        D.synchronize()
        launches += batch
    elapsed = time.perf_counter() - t0
    sampler.stop_flag = True
    sampler.join()
    v = sampler.values[len(sampler.values) // 3:]          # the SMU figure is a moving average: skip the ramp
    return (float(np.mean(v)) if v else float('nan')), launches / elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--json', default='', help='also write the figures here (profiles/pmc_traffic.json merges them)')
    args = ap.parse_args()
    path = busy_file()
    if path is None:
        print('no readable mem_busy_percent under /sys/class/drm: cannot measure HBM-side activity here')
        return 1
    from np_modeling_amd import device as D
    Mat = D.Mat
    rng = np.random.default_rng(0)
    print(f'sampling {path} every 10 ms; each line >= {args.seconds:.0f} s of one kernel in a loop')

    out = {}
    # ---- calibration
    big = [D.empty([1 << 28]) for _ in range(3)]            # 1 GiB each
    for t in big[:2]:
        t.flat_view(0, [1 << 20]).set(rng.standard_normal(1 << 20, dtype=np.float32))
    busy_a, rate_a = measure(path, lambda: D.add(big[0], big[1], out=big[2]), D, args.seconds)
    tbps_a = rate_a * 12.0 * (1 << 28) / 1e12
    per_tbps = busy_a / tbps_a
    if busy_a < 1.0:
        print(f'calibration A (add, 3 x 1 GiB, HBM): {tbps_a:.2f} TB/s of algorithmic traffic at mem_busy {busy_a:.1f} %')
        print('mem_busy_percent stays at 0 under a kernel that streams from HBM at that rate: the SMU figure is not wired up for '
              'this (virtualised) device, and rocprofv3 exposes no memory-controller or Infinity-Cache counters on gfx950 '
              '(TCC / TCP / SQ / SPI / TA / TD / CPC / GRBM only; TCC_EA0_RDREQ_DRAM counts the same requests as TCC_EA0_RDREQ: '
              'profiles/*_pmc_tcc_ffn_dw.log).  HBM-side traffic cannot be separated from Infinity-Cache service here.')
        if args.json:
            with open(args.json, 'w') as f:
                json.dump({'usable': False, 'hbm_stream_tbps': tbps_a, 'hbm_stream_busy_pct': busy_a, 'source': path}, f, indent=1)
        return 0
    print(f'calibration A (add, 3 x 1 GiB, HBM): {tbps_a:.2f} TB/s of algorithmic traffic at mem_busy {busy_a:.1f} % -> {per_tbps:.2f} % per TB/s')
    del big
    small = [D.empty([1 << 22]) for _ in range(3)]          # 16 MiB each: Infinity-Cache resident
    busy_b, rate_b = measure(path, lambda: D.add(small[0], small[1], out=small[2]), D, args.seconds)
    tbps_b = rate_b * 12.0 * (1 << 22) / 1e12
    print(f'calibration B (add, 3 x 16 MiB, Infinity-Cache resident): {tbps_b:.2f} TB/s through L2 at mem_busy {busy_b:.1f} % '
          f'-> the figure {"EXCLUDES" if busy_b < 0.25 * per_tbps * tbps_b else "does NOT exclude"} Infinity-Cache hits')
    del small
    out['calibration'] = {'hbm_stream_tbps': tbps_a, 'hbm_stream_busy_pct': busy_a, 'busy_pct_per_tbps': per_tbps,
                          'mall_resident_tbps': tbps_b, 'mall_resident_busy_pct': busy_b, 'source': path}

    # ---- the GEMM shapes of the C5 step (per-GPU batch 256)
    B, S, F, U = 256, 512, 1024, 4096
    M = B * S

    def buf(n):
        t = D.empty([n])
        t.flat_view(0, [min(n, 1 << 22)]).set(rng.standard_normal(min(n, 1 << 22), dtype=np.float32))
        return t

    x, hbuf, w_ff, w_sq = buf(M * F), buf(M * U), buf(F * U), buf(F * F)
    out_f, out_u, dw = D.empty([M * F]), D.empty([M * U]), D.empty([F * U])
    shapes = {
        'ffn_dw_TN M=1024 N=4096 K=131072': (4.0 * (M * F + M * U + F * U), lambda: D.gemm(F, U, M, Mat(x, F), Mat(hbuf, U), Mat(dw, U), trans_a=True)),
        'ffn_dw_TN M=4096 N=1024 K=131072': (4.0 * (M * F + M * U + F * U), lambda: D.gemm(U, F, M, Mat(hbuf, U), Mat(x, F), Mat(dw, F), trans_a=True)),
        'dw_TN     M=1024 N=1024 K=131072': (4.0 * (2 * M * F + F * F), lambda: D.gemm(F, F, M, Mat(x, F), Mat(out_f, F), Mat(dw, F), trans_a=True)),
        'ffn1_NN   M=131072 N=4096 K=1024': (4.0 * (M * F + F * U + M * U), lambda: D.gemm(M, U, F, Mat(x, F), Mat(w_ff, U), Mat(out_u, U))),
        'ffn2_NN   M=131072 N=1024 K=4096': (4.0 * (M * U + F * U + M * F), lambda: D.gemm(M, F, U, Mat(hbuf, U), Mat(w_ff, F), Mat(out_f, F))),
        'ffn_dx_NT M=131072 N=1024 K=4096': (4.0 * (M * U + F * U + M * F), lambda: D.gemm(M, F, U, Mat(hbuf, U), Mat(w_ff, U), Mat(out_f, F), trans_b=True)),
        'proj_NT   M=131072 N=1024 K=1024': (4.0 * (2 * M * F + F * F), lambda: D.gemm(M, F, F, Mat(x, F), Mat(w_sq, F), Mat(out_f, F), trans_b=True)),
    }
    print(f'{"shape":38s} {"ms/launch":>9s} {"mem_busy":>8s} {"HBM-side GB/launch":>19s} {"algorithmic GB":>15s} {"ratio":>6s}')
    out['gemm'] = {}
    for name, (alg_bytes, fn) in shapes.items():
        busy, rate = measure(path, fn, D, args.seconds)
        ms = 1e3 / rate
        hbm_bytes = busy / per_tbps * 1e12 * (ms * 1e-3)
        print(f'{name:38s} {ms:9.3f} {busy:7.1f}% {hbm_bytes / 1e9:19.2f} {alg_bytes / 1e9:15.2f} {hbm_bytes / alg_bytes:6.2f}')
        out['gemm'][name] = {'ms_per_launch': ms, 'mem_busy_pct': busy, 'hbm_side_bytes_per_launch': hbm_bytes,
                             'algorithmic_bytes_per_launch': alg_bytes, 'ratio': hbm_bytes / alg_bytes}
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(out, f, indent=1)
    return 0


if __name__ == '__main__':
    sys.exit(main())
