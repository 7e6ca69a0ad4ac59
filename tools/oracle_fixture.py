"""A stand-in for scripts/dump_python_parity.py's JSON fixture, produced by the ORACLE (not by upstream Python): same schema
(internal/native/python_parity_test.go:14-38), same inputs.  Used to exercise tests/test_real_checkpoint.py's fixture path on a
synthetic full-size checkpoint; a fixture made this way pins nothing (oracle vs itself) -- only HIP vs fixture is a real comparison.
    python tools/oracle_fixture.py <checkpoint.safetensors> <out.json>"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O


def det(shape, scale):
    n = int(np.prod(shape))
    return (((np.arange(n) % 23) - 11) * scale).astype(np.float32).reshape(shape)


def tj(a):
    a = np.asarray(a, np.float32)
    return {"shape": list(a.shape), "data": [float(x) for x in a.reshape(-1)]}


om = O.OracleModel.from_file(sys.argv[1])
tokens = [10, 20, 30]
st = om.new_state()
om.prompt(st, om.text_embeddings(tokens))
prompt_offsets = [st.offset(i) for i in range(om.n_layers)]
step_latent = det((1, 1, om.ldim), 0.05)
_, _, logit, last = om.step(st, step_latent.reshape(-1), eos_threshold=1e30)
fx = {"flow_lm_prefill_step": {"tokens": tokens, "step_latent": tj(step_latent), "prompt_layer_offsets": prompt_offsets,
                               "step_layer_offsets": [st.offset(i) for i in range(om.n_layers)],
                               "step_last_hidden": tj(np.asarray(last).reshape(1, -1)), "step_eos_logits": tj(np.array([[logit]], np.float32))},
      "mimi": []}
for frames in (1, 2, 4):
    lat = det((1, frames, om.ldim), 0.03)
    mimi = om.latent_to_mimi(lat)
    fx["mimi"].append({"name": f"{frames}_frames", "latent": tj(lat), "latent_to_mimi": tj(mimi), "mimi_decode": tj(om.mimi_decode(mimi))})
json.dump(fx, open(sys.argv[2], "w"))
print("fixture written:", sys.argv[2])
