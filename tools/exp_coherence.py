"""Experiment: extend-kernel throughput on the dragon stand-in for differently ordered ray sets."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G, scenes as S
pkg = G.load_package()
W, H = 1920, 1080
scene = pkg.HostScene(S.scene_file(pkg, "dragon"), W, H)
ctx = pkg.Context(0); ctx.upload(scene)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx.set_option(pkg.OPT_EXTEND_MODE, mode)
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
parts = [ctx.primary_rays(W, H, sample=s_, seed=1) for s_ in range(NS)]
# put rays in the generate stage's tile order: 8x4 tiles
ys, xs = np.divmod(np.arange(W * H), W)
def tile_order():
    r = np.arange(W * H); tile = r >> 5; inn = r & 31; tpr = W >> 3
    ty = tile // tpr; tx = tile - ty * tpr
    x = tx * 8 + (inn & 7); y = ty * 4 + (inn >> 3)
    return (H - y - 1) * W + x
perm = tile_order()
org = np.concatenate([p_[0][perm] for p_ in parts]); dir_ = np.concatenate([p_[1][perm] for p_ in parts])
def run(o, d, label):
    n = o.shape[0]
    o4 = torch.zeros((n, 4)); d4 = torch.zeros((n, 4)); o4[:, :3] = torch.from_numpy(o); d4[:, :3] = torch.from_numpy(d)
    o4 = o4.cuda(); d4 = d4.cuda(); hit = torch.empty((n, 4), device="cuda"); t = torch.empty((n,), device="cuda")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        st = ctx.intersect_device(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
        best = min(best, st.ms_total)
    print("%-40s n=%8d  %7.3f ms  %8.1f Mrays/s  hit %.2f" % (label, n, best, n / best / 1e3, float((t < 1e38).float().mean())))
    return hit.cpu().numpy(), t.cpu().numpy()
o_t, d_t = org, dir_
hit, t = run(o_t, d_t, "primary, tile order")
rng = np.random.default_rng(0)
sh = rng.permutation(org.shape[0])
run(org[sh], dir_[sh], "primary, shuffled")
# secondary rays: cosine-ish around the normal at the hit
nrm = hit[:, :3]; m = t < 1e38
o2, d2 = S.secondary_rays(o_t, d_t, t, nrm, 5)
run(o2, d2, "secondary, path order")
sh2 = rng.permutation(o2.shape[0])
run(o2[sh2], d2[sh2], "secondary, shuffled")
# secondary sorted by origin morton code (10 bits per axis)
lo = o2.min(0); hi = o2.max(0); q = np.clip(((o2 - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint32), 0, 1023)
def part(v):
    v = v.astype(np.uint64); v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
code = part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2)
octant = ((d2[:, 0] > 0).astype(np.uint64) | ((d2[:, 1] > 0).astype(np.uint64) << 1) | ((d2[:, 2] > 0).astype(np.uint64) << 2))
srt = np.argsort(code, kind="stable")
run(o2[srt], d2[srt], "secondary, sorted by origin morton")
srt2 = np.argsort((octant << 30) | code, kind="stable")
run(o2[srt2], d2[srt2], "secondary, sorted by octant+morton")
