"""Experiment (needs libvar_TRIV.so built with -DPT_VARIANT_COUNT_TRIVIAL): share of rays whose traversal ends at the root."""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["PT_HIP_LIB"] = "/root/repo/cuda-path-tracer_amd/libvar_TRIV.so"
import __graft_entry__ as G, scenes as S
pkg = G.load_package()
W, H, spp = 960, 540, 16
for name in ("dragon", "bunny", "city"):
    scene = pkg.HostScene(S.scene_file(pkg, name), W, H)
    c = pkg.Context(0); c.set_option(pkg.OPT_COUNT_TRAVERSAL, 1); c.upload(scene); c.film_bind(W, H)
    st = c.render(0, spp, spp, seed=1)
    print("%s: %.1f %% of %d rays visit at most the root node (%.1f %% of all rays do so and hit nothing); %.2f nodes per ray; paths %d"
          % (name, 100.0 * st.leaves / st.rays, st.rays, 100.0 * st.tris / st.rays, st.nodes / st.rays, st.paths))
    c.close()
