import sys, time, os
sys.path.insert(0,'.')
import torch
from oracle import tris_oracle as O
from tris_amd.utils.shapes import aux_state_dict_spec, empty_state_dict, tris_state_dict_spec
from tris_amd.utils.synth import seed_fill, synthetic_batch
thr=int(sys.argv[1]); B=int(sys.argv[2])
torch.set_num_threads(thr)
sd = seed_fill(empty_state_dict(tris_state_dict_spec()), 1234)
aux = seed_fill(empty_state_dict(aux_state_dict_spec()), 4321)
b = synthetic_batch(B, 320, 20, 3, seed=7)
st={}
for i in range(3):
    t0=time.perf_counter(); O.train_step(sd, aux, b, state=st, faithful=True); dt=time.perf_counter()-t0
    print(thr, B, i, f"{dt:.2f}s  {B/dt:.3f} img/s", flush=True)
