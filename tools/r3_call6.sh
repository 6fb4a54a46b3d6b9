#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "recorded_training or adam" ) > gpurun_out/r3/t6.log 2>&1
tail -30 gpurun_out/r3/t6.log
python - <<'PY'
import time, torch, sys
sys.path.insert(0,'.')
from ihgnn_amd import synth
from ihgnn_amd.Dataset import GraphDataset
from ihgnn_amd.Models import HemPredictionLayer, IHGNNLayer, RawGnn
from ihgnn_amd.optim import Adam
from ihgnn_amd.captured_step import CapturedTrainingStep
dev=torch.device('cuda:0')
for cfg in ('C1','C2','C3'):
    c=synth.CONFIGS[cfg]; w=synth.draw_config(cfg)
    ds=GraphDataset.from_arrays(w.user_count,w.query_count,w.item_count,w.vocab_size,w.bag_words,w.bag_offsets,w.triples,device=dev)
    torch.manual_seed(0)
    m=RawGnn(dev,ds,c['dim'],IHGNNLayer,c['layers'],3,False,HemPredictionLayer,0.5).to(dev)
    m.batch_rows_only_last_layer=False
    opt=Adam(m.parameters(),1e-3,weight_decay=0)
    b=list(ds.sample_batches(100,40,seed=1))
    for k in range(5):
        m.bce_loss(*b[k]).backward(); opt.step(); opt.zero_grad()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for k in range(5,35):
        m.bce_loss(*b[k]).backward(); opt.step(); opt.zero_grad()
    torch.cuda.synchronize(); eager=(time.perf_counter()-t0)/30*1e3
    st=CapturedTrainingStep(m,opt,1100,warmup_batch=b[0])
    for k in range(3): st.step(*b[k])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for k in range(5,35): st.step(*b[k])
    torch.cuda.synchronize(); rec=(time.perf_counter()-t0)/30*1e3
    print(cfg,'eager ms',round(eager,4),'recorded ms',round(rec,4), flush=True)
    del m,opt,st,ds
PY
