import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G, oracle_lib as O, scenes as S
pkg = G.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "suzanne"
scene = S.parity_cases(pkg)[name]
g = np.load(os.path.join(S.GOLDEN, "rays_%s.npz" % name))
ctx = pkg.Context(0); ctx.upload(scene)
t, prim, _ = ctx.intersect(g["org"], g["dir"])
bad = np.nonzero((prim != g["prim"]) | (t != g["t"]))[0]
print("bad", bad)
for i in bad[:10]:
    print(i, "org", g["org"][i].tolist(), [hex(x) for x in g["org"][i].view(np.uint32)], "dir", g["dir"][i].tolist(), [hex(x) for x in g["dir"][i].view(np.uint32)])
    print("   gpu", t[i], hex(prim[i]), "oracle", g["t"][i], hex(g["prim"][i]))
