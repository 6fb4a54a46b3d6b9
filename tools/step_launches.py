#!/usr/bin/env python3
"""A few full training steps of a small d = 128 model (eager, then replayed from a recording), for a kernel trace of what a step launches:

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/step_launches.py [--recorded]

tests/test_gpu_parity.py::test_training_step_launches_no_framework_kernels cuts the trace at the Adam launches and checks that nothing between two of them
is a framework kernel (at::native::*, __amd_rocclr_*)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ihgnn_amd import ops, synth
from ihgnn_amd.Dataset import GraphDataset
from ihgnn_amd.Models import HemPredictionLayer, IHGNNLayer, RawGnn
from ihgnn_amd.optim import Adam


def main():
    dev = torch.device('cuda:0')
    w = synth.draw(3000, 400, 2000, 200, 40000, seed=3, distribution='powerlaw')
    ds = GraphDataset.from_arrays(w.user_count, w.query_count, w.item_count, w.vocab_size, w.bag_words, w.bag_offsets, w.triples, device=dev)
    torch.manual_seed(0)
    model = RawGnn(dev, ds, 128, IHGNNLayer, 3, 3, False, HemPredictionLayer, 0.5).to(dev)
    model.batch_rows_only_last_layer = False
    opt = Adam(model.parameters(), 1e-3, weight_decay=0)
    batches = list(ds.sample_batches(100, 8, seed=1))
    if '--recorded' in sys.argv:
        from ihgnn_amd.captured_step import CapturedTrainingStep
        step = CapturedTrainingStep(model, opt, int(batches[0][0].shape[0]), warmup_batch=batches[0])
        for b in batches:
            step.step(*b)
    else:
        for u, q, i, y in batches:
            loss = model.bce_loss(u, q, i, y)
            ops.backward(loss)
            opt.step()
            opt.zero_grad()
    torch.cuda.synchronize()
    print('steps done')


if __name__ == '__main__':
    main()
