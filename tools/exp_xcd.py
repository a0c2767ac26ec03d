"""Experiment: does giving each XCD a contiguous share of the ray buffer (PT_OPT_TUNE_XCD_GROUP) help the extend kernel,
for rays in path order and for rays sorted by the Morton code of their origin?  Dragon stand-in, one-bounce rays."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G, scenes as S
pkg = G.load_package()
W, H = 1920, 1080
scene = pkg.HostScene(S.scene_file(pkg, "dragon"), W, H)
ctx = pkg.Context(0); ctx.upload(scene)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
parts = [ctx.primary_rays(W, H, sample=s_, seed=1) for s_ in range(NS)]
r = np.arange(W * H); tile = r >> 5; inn = r & 31; tpr = W >> 3
ty = tile // tpr; tx = tile - ty * tpr
perm = (H - (ty * 4 + (inn >> 3)) - 1) * W + tx * 8 + (inn & 7)
org = np.concatenate([p_[0][perm] for p_ in parts]); dir_ = np.concatenate([p_[1][perm] for p_ in parts])

def run(o, d, label, reps=4):
    n = o.shape[0]
    o4 = torch.zeros((n, 4)); d4 = torch.zeros((n, 4)); o4[:, :3] = torch.from_numpy(o); d4[:, :3] = torch.from_numpy(d)
    o4 = o4.cuda(); d4 = d4.cuda(); hit = torch.empty((n, 4), device="cuda"); t = torch.empty((n,), device="cuda")
    torch.cuda.synchronize()
    out = []
    for g in (0, 1):
        ctx.set_option(pkg.OPT_TUNE_XCD_GROUP, g)
        best = min(ctx.intersect_device(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr()).ms_total for _ in range(reps))
        out.append(best)
    print("%-36s n=%9d  plain %7.3f ms %8.1f Mrays/s | xcd-grouped %7.3f ms %8.1f Mrays/s" %
          (label, n, out[0], n / out[0] / 1e3, out[1], n / out[1] / 1e3), flush=True)
    return hit.cpu().numpy(), t.cpu().numpy()

hit, t = run(org, dir_, "primary, tile order")
o2, d2 = S.secondary_rays(org, dir_, t, hit[:, :3], 5)
hit2, t2 = run(o2, d2, "one bounce, path order")
lo = o2.min(0); hi = o2.max(0); q = np.clip(((o2 - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint32), 0, 1023)
def part(v):
    v = v.astype(np.uint64); v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
code = part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2)
srt = np.argsort(code, kind="stable")
run(o2[srt], d2[srt], "one bounce, sorted by origin morton")
# coarse bins only (what a binned append could give): 6 Morton bits = 64 cells, path order inside a cell
srt6 = np.argsort(code >> 24, kind="stable")
run(o2[srt6], d2[srt6], "one bounce, 64 origin cells")
o3, d3 = S.secondary_rays(o2, d2, t2, hit2[:, :3], 6)
run(o3, d3, "two bounces, path order")
lo = o3.min(0); hi = o3.max(0); q = np.clip(((o3 - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint32), 0, 1023)
code = part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2)
run(o3[np.argsort(code, kind="stable")], d3[np.argsort(code, kind="stable")], "two bounces, sorted by origin morton")
