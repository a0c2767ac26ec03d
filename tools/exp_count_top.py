import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as G, scenes as S
pkg = G.load_package()
W, H, spp = 960, 540, 16
scene = pkg.HostScene(S.scene_file(pkg, "dragon"), W, H)
for k in (7, 15, 31, 63, 127):
    c = pkg.Context(0); c.set_option(10, k); c.set_option(pkg.OPT_COUNT_TRAVERSAL, 1); c.upload(scene); c.film_bind(W, H)
    st = c.render(0, spp, spp, seed=1)
    print("top %3d nodes: %.3f of %.3f node visits per ray (%.1f %%)" % (k, st.tris / st.rays, st.nodes / st.rays, 100.0 * st.tris / st.nodes))
    c.close()
