"""Generate tests/golden/gae_seed23.npz.

The inputs are those of the reference's known-answer GAE test
(`nnx_ppo/algorithms/ppo_test.py:229-264`: np.random.seed(23), T=100, N=512,
gamma=0.8, lambda=0.95, P(done)=0.01, truncation subset of done) and the
expected output is that test's ground-truth loop, both restated in
`oracle/gae.py`.  Run from the repo root:  python tests/golden/make_gae_seed23.py
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.gae import gae_known_answer, make_seed23_case  # noqa: E402

c = make_seed23_case()
adv = gae_known_answer(c["rewards"], c["values"], c["done"], c["truncation"],
                       c["gamma"], c["lambda_"])
out = Path(__file__).with_name("gae_seed23.npz")
np.savez_compressed(
    out, rewards=c["rewards"], values=c["values"], done=c["done"],
    truncation=c["truncation"], gamma=c["gamma"], lambda_=c["lambda_"],
    advantages=adv)
a32 = adv.astype(np.float32)
print(out, adv.shape, "mean %.9f std %.9f" % (adv.mean(), adv.std()),
      "sum(done)=%d sum(trunc)=%d" % (c["done"].sum(), c["truncation"].sum()),
      "sha256(f32)[:16]=" + hashlib.sha256(a32.tobytes()).hexdigest()[:16])
