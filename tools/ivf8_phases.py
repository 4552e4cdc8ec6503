#!/usr/bin/env python3
"""Phase times of the search filter's persistent workgroups (a build with -DGNNLM_IVF8_EXP=512; GNNLM_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnnlm_amd.synthetic import synthetic_ivfpq_index

dev = torch.device("cuda:0")
idx = synthetic_ivfpq_index(103227021, 1024, 4096, 64, dev)
idx.keep_work_ctr = True
torch.manual_seed(0)
q = torch.randn(8192, 1024, device=dev); q = q / q.norm(dim=1, keepdim=True)
for _ in range(2):
    idx.search_device(q, 1024); torch.cuda.synchronize()
c = idx.last_work_ctr.cpu().double()          # the last scan8 call = the filter
names = ["set-up + table fill", "tile loop", "records out (+ loop top)", "own totals", "wait for the other waves", "counters (atomics)"]
tot = c[:, 1:7].sum()
for i, nm in enumerate(names):
    print("%-28s %5.1f %%" % (nm, 100 * c[:, 1 + i].sum() / tot))
print("ticks per workgroup %.0f" % (tot / 256))
n = c[:, 11].sum()
print("wave 8, ticks per tile: look-ups + matrix instructions %.0f, flush check %.0f, compares + appends %.0f  (tiles %.0f)" % (
    c[:, 8].sum() / n, c[:, 9].sum() / n, c[:, 10].sum() / n, n))
print("thread 0 loop ticks per tile: %.0f" % (c[:, 2].sum() / n))
print("shader clock over the workgroups' lifetimes: %.0f MHz (clock64 / wall_clock64 at 100 MHz); lifetime %.2f ms" % (
    16 * c[:, 12].sum() / c[:, 13].sum() * 100, c[:, 13].sum() / 256 / 1e5))
