import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from ld_tools_amd import ld_from_counts
from oracle import c_oracle
n = 5008
rng = np.random.RandomState(n)
m = 2_000_000
kind = rng.randint(0, 6, m)
u = rng.rand(m)
a1 = np.where(kind == 1, rng.randint(0, 4, m), (np.sin(np.pi / 2 * u) ** 2 * n).astype(np.int64))
a2 = np.where(kind == 2, rng.randint(0, 4, m), (np.sin(np.pi / 2 * rng.rand(m)) ** 2 * n).astype(np.int64))
miss1 = np.where(kind == 3, rng.randint(0, n // 3, m), 0)
miss2 = np.where(kind == 3, rng.randint(0, n // 3, m), 0)
a1 = np.minimum(a1, n - miss1)
a2 = np.minimum(a2, n - miss2)
r1, r2 = n - miss1 - a1, n - miss2 - a2
lo = np.maximum(0, a1 + a2 - n)
hi = np.minimum(a1, a2)
indep = np.rint(a1.astype(np.float64) * a2 / n).astype(np.int64)
n11 = np.where(kind == 4, indep + rng.randint(-2, 3, m), lo + (rng.rand(m) * (hi - lo + 1)).astype(np.int64))
n11 = np.where(kind == 5, rng.choice([0, 1], m) * hi + (1 - rng.choice([0, 1], m)) * lo, n11)
n11 = np.clip(n11, np.where(kind == 3, 0, lo), hi)
arrs = [x.astype(np.uint32) for x in (n11, a1, r1, a2, r2)]
raw, rnd, flags = ld_from_counts(n, *arrs)
rnd = rnd.cpu().numpy(); raw = raw.cpu().numpy()
o_rsq_raw, o_dp_raw, o_rsq, o_dp, o_flags = c_oracle.ld_from_counts_v(n, *arrs, libm_pow=True)
k = np.rint(rnd[:, 1].astype(np.float64) * 1e4).astype(np.int64)
ko = np.rint(o_dp * 1e4).astype(np.int64)
bad = np.nonzero((k != ko) & (o_dp < 1600))[0]
print("mismatches:", len(bad))
for i in bad[:12]:
    print(dict(kind=int(kind[i]), n11=int(n11[i]), a1=int(a1[i]), r1=int(r1[i]), a2=int(a2[i]), r2=int(r2[i]), gpu_k=int(k[i]), oracle_k=int(ko[i]),
               oracle_dp=float(o_dp[i]), oracle_dp_raw=repr(float(o_dp_raw[i])), gpu_raw=repr(float(raw[i, 1]))))
