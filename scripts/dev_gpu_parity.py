import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from slam_amd import ekf, synth
import helpers
print('mfma probe TF/s:', ekf.mfma_f64_probe(20000), flush=True)
st = synth.gen_stream(seed=1, nf=20, steps=31, churn=True)
so, f = helpers.run_oracle(st)
fb = ekf.FilterBatch(max_features=24, batch=1)
sg = helpers.run_gpu(st, fb)
for k,(a,b) in enumerate(zip(so, sg)):
    print(k, 'ids', a['ids']==b['ids'], 'X', helpers.rel_err(b['X'], a['X']), 'V', helpers.rel_err(b['V'],a['V']), 'pib', helpers.rel_err(b['pib'],a['pib']), 'P', helpers.rel_err(b['P'],a['P']), 'applied', a['applied'], b['applied'], flush=True)
