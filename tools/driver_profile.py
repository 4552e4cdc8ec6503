#!/usr/bin/env python3
"""Where the drop-in driver (`eval_lm.main` -> `SequenceScorer.generate`) spends a batch beyond the kernels of the step:
torch.profiler table of one bench-sized batch (32 blocks) on the bench's resident tables."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

if __name__ == "__main__":
    sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-parity", "--no-extras"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    eng, shard, sharded, cpu_model, (d, vocab) = bench.build(args, dev, 0, 1)
    batches = bench.make_batches(args, dev, 0, d, vocab)
    bench.driver_path(args, eng, batches, dev)                      # warm
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = bench.driver_path(args, eng, batches, dev)
        torch.cuda.synchronize()
    print(out)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=15, max_name_column_width=60))
