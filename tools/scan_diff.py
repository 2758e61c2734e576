"""debug helper: GPU scan m[] vs the oracle on one golden case (usage: scan_diff.py case [nbytes])"""
import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib, golden_util
from x3_compressor_amd import _lib
cases = golden_util.load_cases()
c = cases[sys.argv[1]]
data = c['data'][:int(sys.argv[2])] if len(sys.argv) > 2 else c['data']
prm = _lib.params_from_args(c['args'])
ctx = _lib.X3Context(0)
m_gpu = np.asarray(ctx.scan_m(data, prm))
o = oracle_lib.load()
a = c['args']
kw = {}
for i, x in enumerate(a):
    if x == '-w': kw['w_kib'] = int(a[i + 1])
    if x == '-t': kw['t'] = int(a[i + 1])
m_or = np.asarray(o.scan_m(data, oracle_lib.params(**kw)))
d = np.nonzero(m_gpu != m_or)[0]
print(len(data), "diffs", len(d), d[:40], m_gpu[d[:40]], m_or[d[:40]])
for p in d[:10]:
    print(p, bytes(data[p:p + 8]))
