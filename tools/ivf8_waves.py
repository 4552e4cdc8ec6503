#!/usr/bin/env python3
"""When do the 16 waves of the search filter's workgroups leave a group's tile loop?  (a build with -DGNNLM_IVF8_EXP=1024; GNNLM_LIB=...)"""
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
c = idx.last_work_ctr.cpu().double().sum(0)
t = c[1:15]                                  # waves 1 .. 14 (slot 15 collects waves 0 and 15: their mean below)
both = c[15] / 2
print("mean exit time of wave w relative to the latest of them (w = 1 .. 14, then the mean of waves 0 and 15):",
      " ".join("%.2f" % (x / max(t.max(), both)) for x in list(t) + [both]))
