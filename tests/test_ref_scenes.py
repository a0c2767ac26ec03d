"""The reference's own .scene files, read where they lie (only where /root/reference exists; CPU only).

Two things are held: the scene texts the test-suite and the benchmark carry (tests/scenes.py SCENE_TEXT) are the
reference's files word for word apart from the asset paths, and every one of the six shipped files loads through the host
library as written (the last line of some has no newline; assets the reference does not ship are replaced by the stand-ins)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import scenes as S

REF_SCENES = "/root/reference/scene"
REF_OBJ = "/root/reference/obj"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="the reference's scene files are not on this machine")

ASSET = re.compile(r"^(file_name)\s+(\S+)\s*$", re.M)


def blocks_of(text):
    """[(keyword, {key: [values]})] of a .scene text, asset paths reduced to their file names."""
    out = []
    for chunk in re.split(r"\n\s*\n", text.strip()):
        lines = [ln.split() for ln in chunk.strip().splitlines() if ln.strip()]
        kv = {ln[0]: ([os.path.basename(ln[1])] if ln[0] == "file_name" else ln[1:]) for ln in lines[1:]}
        out.append((lines[0][0], kv))
    return out


@needs_ref
@pytest.mark.parametrize("name", ["balls", "bunny", "dragon", "city", "suzanne"])
def test_carried_scene_texts_are_the_reference_files(name):
    ref = open(os.path.join(REF_SCENES, name + ".scene")).read()
    names = {"studio": "studio.hdr", "sunset": "sunset.hdr", "forest": "forest.hdr", "bunny": "bunny.obj",
             "city": "city.obj", "dragon": "dragon.obj", "suzanne": "suzanne.obj"}
    mine = S.SCENE_TEXT[name].format(**names)
    assert blocks_of(mine) == blocks_of(ref)


@needs_ref
@pytest.mark.parametrize("name", ["balls", "bunny", "dragon", "city", "scream", "suzanne"])
def test_reference_scene_files_load_as_written(pkg, tmp_path, name):
    text = open(os.path.join(REF_SCENES, name + ".scene")).read()

    def resolve(m):
        base = os.path.basename(m.group(2))
        if base.endswith(".hdr"):                       # none of the maps is shipped
            return "file_name " + S.ensure_asset(pkg, base)
        shipped = os.path.join(REF_OBJ, base)
        if os.path.exists(shipped):
            return "file_name " + shipped
        stand_in = {"dragon.obj": "knot_small.obj", "scream.obj": "knot_small.obj"}[base]
        return "file_name " + S.ensure_asset(pkg, stand_in)

    path = tmp_path / (name + ".scene")
    path.write_text(ASSET.sub(resolve, text))           # a file without a final newline stays without
    scene = pkg.HostScene(str(path), 64, 36)
    d = scene.desc
    want = blocks_of(text)
    kinds = [k for k, _ in want]
    assert d.num_planes == kinds.count("PLANE") and d.num_spheres == kinds.count("SPHERE")
    assert (d.num_nodes > 0) == ("OBJ" in kinds)
    types = {"LAMBERTIAN": pkg.DIFFUSE, "MIRROR": pkg.MIRROR, "GLOSSY": pkg.GLOSSY, "GLASS": pkg.GLASS}
    for kind, kv in want:
        if kind == "OBJ":
            assert d.mesh_material_type == types[kv["material_type"][0]]
            assert np.allclose(list(d.mesh_material)[:3], [float(v) for v in kv["albedo"]])
            shipped = os.path.join(REF_OBJ, kv["file_name"][0])
            if os.path.exists(shipped):                 # the triangle count the reference's loader prints for it
                faces = sum(1 for ln in open(shipped) if ln.startswith("f "))
                assert d.num_tris == faces
    dplanes = C.cast(d.planes, C.POINTER(pkg.Plane))
    dspheres = C.cast(d.spheres, C.POINTER(pkg.Sphere))
    planes = [kv for kind, kv in want if kind == "PLANE"]
    for i, kv in enumerate(planes):
        assert dplanes[i].material_type == types[kv["material_type"][0]]
        n = np.array([float(v) for v in kv["normal"]])
        assert np.allclose(list(dplanes[i].normal), n / np.linalg.norm(n), atol=1e-6)
    spheres = [kv for kind, kv in want if kind == "SPHERE"]
    for i, kv in enumerate(spheres):
        assert dspheres[i].material_type == types[kv["material_type"][0]]
        assert abs(dspheres[i].radius - float(kv["radius"][0])) < 1e-6
        w = dspheres[i].material[3]
        if "refraction_index" in kv:
            assert abs(w - float(kv["refraction_index"][0])) < 1e-6
