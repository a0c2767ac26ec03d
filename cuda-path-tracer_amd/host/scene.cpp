// .scene / .obj ingestion and camera set-up (host).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "host_internal.h"

namespace pth {

static thread_local std::string g_error;

int set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return -1;
}

const std::string &last_error() { return g_error; }

// Camera::Camera, reference src/camera.cuh:21-37 (host arithmetic, unfused)
void camera_make(const float pos[3], const float look_at[3], float fov, float aspect, float aperture, float focus_dist,
                 pt_camera *out)
{
    constexpr float kPi = 3.14159265358979323846264338327950288f;
    struct V { float x, y, z; };
    auto sub = [](V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto mul = [](V a, float s) { return V{a.x * s, a.y * s, a.z * s}; };
    auto dot = [](V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto cross = [](V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    auto unit = [&](V v) { return mul(v, 1.0f / sqrtf(dot(v, v))); };
    const float height = 2.0f * tanf(fov * kPi / 360.0f);
    const float width = aspect * height;
    const V origin{pos[0], pos[1], pos[2]};
    const V w = unit(sub(origin, V{look_at[0], look_at[1], look_at[2]}));
    const V u = unit(cross(V{0, 1, 0}, w));
    const V v = cross(w, u);
    const V right = mul(mul(u, focus_dist), width);
    const V up = mul(mul(v, focus_dist), height);
    const V r2{-right.x / 2, -right.y / 2, -right.z / 2};
    const V u2{up.x / 2, up.y / 2, up.z / 2};
    const V wf = mul(w, focus_dist);
    const V bl{r2.x - u2.x - wf.x, r2.y - u2.y - wf.y, r2.z - u2.z - wf.z};
    auto put = [](float *d, V s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    put(out->origin, origin); put(out->bottom_left, bl); put(out->right, right); put(out->up, up);
    put(out->u, u); put(out->v, v); put(out->w, w);
    out->lens_radius = aperture / 2;
}

// Wavefront .obj: `v x y z` and `f a b c` records (reference src/scene.cu:15-55).  Accepted on top of that: comments,
// blank lines, vn/vt/other records (skipped), a/b/c index forms and polygons (fanned).  Index 0 is a dummy vertex as
// in the reference; negative indices are relative to the vertices read so far.
int obj_load(const std::string &path, std::vector<float> &verts, uint64_t &ntris)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return set_error("cannot open %s", path.c_str());
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<float> pos(3, 0.0f);
    verts.clear();
    const char *p = text.c_str(), *end = p + text.size();
    while (p < end) {
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        while (p < eol && (*p == ' ' || *p == '\t')) p++;
        if (p < eol && p[0] == 'v' && p + 1 < eol && (p[1] == ' ' || p[1] == '\t')) {
            // three coordinates, all on this line: strtof skips newlines, so a short record must not be allowed to take
            // the next record's first number
            char *q = const_cast<char *>(p + 1);
            for (int k = 0; k < 3; k++) {
                while (q < eol && (*q == ' ' || *q == '\t')) q++;
                char *e = q;
                const float c = q < eol ? std::strtof(q, &e) : 0.0f;
                if (e == q || e > eol) return set_error("%s: vertex record with fewer than 3 coordinates", path.c_str());
                // an infinite or NaN coordinate would poison every box above it in the tree
                if (!std::isfinite(c)) return set_error("%s: vertex coordinate is not a finite number", path.c_str());
                pos.push_back(c);
                q = e;
            }
        } else if (p < eol && p[0] == 'f' && p + 1 < eol && (p[1] == ' ' || p[1] == '\t')) {
            char *q = const_cast<char *>(p + 1);
            long idx[3];
            int have = 0;
            for (;;) {
                while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
                if (q >= eol) break;
                char *e;
                long i = std::strtol(q, &e, 10);
                if (e == q) break;
                q = e;
                while (q < eol && *q != ' ' && *q != '\t') q++;      // skip /vt/vn
                const long nverts = (long)(pos.size() / 3);
                if (i < 0) i += nverts;
                if (i < 0 || i >= nverts) return set_error("%s: face index out of range", path.c_str());
                if (have < 3) idx[have++] = i;
                else { idx[1] = idx[2]; idx[2] = i; }
                if (have == 3) {
                    for (int k = 0; k < 3; k++) {
                        const float *v = &pos[3 * (size_t)idx[k]];
                        verts.insert(verts.end(), v, v + 3);
                    }
                }
            }
            if (have < 3) return set_error("%s: face record with fewer than 3 vertices", path.c_str());
        }
        p = eol + 1;
    }
    ntris = verts.size() / 9;
    return 0;
}

namespace {

struct Block {
    std::string kind;
    std::map<std::string, std::vector<std::string>> kv;
};

// A .scene file is a sequence of blocks: a keyword line (CAMERA, ENVIRONMENT, SPHERE, PLANE, OBJ) followed by
// `key value...` lines (reference src/scene.cu:146-201, 96-136).  A blank line or the next keyword ends a block.
std::vector<Block> read_blocks(std::istream &in)
{
    std::vector<Block> blocks;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream iss(line);
        std::string key;
        iss >> key;
        if (key.empty()) continue;
        const bool keyword = key == "CAMERA" || key == "ENVIRONMENT" || key == "SPHERE" || key == "PLANE" || key == "OBJ";
        if (keyword) {
            blocks.push_back(Block{key, {}});
            continue;
        }
        if (blocks.empty()) continue;       // unknown text before the first block: ignored like the reference does
        std::vector<std::string> vals;
        std::string v;
        while (iss >> v) vals.push_back(v);
        blocks.back().kv[key] = vals;
    }
    return blocks;
}

bool get_floats(const Block &b, const char *key, float *out, int n)
{
    auto it = b.kv.find(key);
    if (it == b.kv.end() || (int)it->second.size() < n) return false;
    for (int i = 0; i < n; i++) out[i] = std::strtof(it->second[i].c_str(), nullptr);
    return true;
}

bool get_string(const Block &b, const char *key, std::string &out)
{
    auto it = b.kv.find(key);
    if (it == b.kv.end() || it->second.empty()) return false;
    out = it->second[0];
    return true;
}

// reference src/scene.cu:96-136
int read_material(const Block &b, float mat[4], uint32_t &type)
{
    std::string mt;
    if (!get_string(b, "material_type", mt)) return set_error("%s block without material_type", b.kind.c_str());
    float albedo[3];
    if (!get_floats(b, "albedo", albedo, 3)) return set_error("%s block without albedo", b.kind.c_str());
    mat[0] = albedo[0]; mat[1] = albedo[1]; mat[2] = albedo[2]; mat[3] = 0;
    if (mt == "LAMBERTIAN") type = PT_DIFFUSE;
    else if (mt == "MIRROR") type = PT_MIRROR;
    else if (mt == "GLASS") {
        if (!get_floats(b, "refraction_index", &mat[3], 1)) return set_error("GLASS without refraction_index");
        type = PT_GLASS;
    } else if (mt == "GLOSSY") {
        if (!get_floats(b, "glossyness", &mat[3], 1)) return set_error("GLOSSY without glossyness");
        type = PT_GLOSSY;
    } else {
        return set_error("Invalid material type: %s", mt.c_str());
    }
    return 0;
}

std::string resolve(const std::string &name, const std::string &scene_file)
{
    if (std::ifstream(name).good()) return name;        // relative to the working directory, as the reference
    const size_t slash = scene_file.find_last_of('/');
    if (slash != std::string::npos) {
        const std::string alt = scene_file.substr(0, slash + 1) + name;
        if (std::ifstream(alt).good()) return alt;
    }
    return name;
}

}   // namespace
}   // namespace pth

using namespace pth;

extern "C" {

const char *pth_last_error(void) { return pth::last_error().c_str(); }

int pth_scene_load(const char *scene_file, uint32_t width, uint32_t height, pth_scene **out)
{
    if (!scene_file || !out || width < 2 || height < 2) return set_error("pth_scene_load: bad argument");
    *out = nullptr;
    std::ifstream in(scene_file);
    if (!in) return set_error("cannot open %s", scene_file);
    const std::vector<Block> blocks = read_blocks(in);
    pth_scene *sc = new pth_scene();
    bool have_camera = false, have_env = false;
    std::vector<float> all_verts;           // triangles of all OBJ blocks, in block order
    int rc = 0;
    for (const Block &b : blocks) {
        if (b.kind == "CAMERA") {
            if (have_camera) { rc = set_error("only one camera allowed per scene"); break; }
            float pos[3], look[3], fov, aperture, focus;
            if (!get_floats(b, "position", pos, 3) || !get_floats(b, "look_at", look, 3) || !get_floats(b, "fov", &fov, 1) ||
                !get_floats(b, "aperture", &aperture, 1) || !get_floats(b, "focus_dist", &focus, 1)) {
                rc = set_error("CAMERA block incomplete");
                break;
            }
            camera_make(pos, look, fov, static_cast<float>(width) / height, aperture, focus, &sc->desc.camera);
            have_camera = true;
        } else if (b.kind == "ENVIRONMENT") {
            if (have_env) { rc = set_error("only one environment allowed per scene"); break; }
            std::string name;
            if (!get_string(b, "file_name", name)) { rc = set_error("ENVIRONMENT without file_name"); break; }
            uint32_t w = 0, h = 0;
            std::cout << "Building environment map...\n";
            if ((rc = hdr_load(resolve(name, scene_file), sc->env, w, h)) != 0) break;
            sc->marginal.resize(w);
            sc->conditional.resize((size_t)w * h);
            if ((rc = env_build(sc->env.data(), w, h, sc->marginal.data(), sc->conditional.data())) != 0) break;
            sc->desc.env_width = w;
            sc->desc.env_height = h;
            have_env = true;
        } else if (b.kind == "SPHERE") {
            pt_sphere s{};
            if ((rc = read_material(b, s.material, s.material_type)) != 0) break;
            if (!get_floats(b, "position", s.position, 3) || !get_floats(b, "radius", &s.radius, 1)) {
                rc = set_error("SPHERE block incomplete");
                break;
            }
            sc->spheres.push_back(s);
        } else if (b.kind == "PLANE") {
            pt_plane p{};
            if ((rc = read_material(b, p.material, p.material_type)) != 0) break;
            float n[3];
            if (!get_floats(b, "position", p.position, 3) || !get_floats(b, "normal", n, 3)) {
                rc = set_error("PLANE block incomplete");
                break;
            }
            // Plane's constructor normalises (reference src/object.cuh:55; cutil normalize = v * (1/sqrt(dot)))
            const float s = 1.0f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            p.normal[0] = n[0] * s; p.normal[1] = n[1] * s; p.normal[2] = n[2] * s;
            sc->planes.push_back(p);
        } else if (b.kind == "OBJ") {
            // The reference allows one OBJ block (src/scene.cu:187-188).  Several are accepted here: their triangles go
            // into ONE BVH in block order, each mesh keeping its own material.
            pt_mesh m{};
            if ((rc = read_material(b, m.material, m.material_type)) != 0) break;
            std::string name;
            if (!get_string(b, "file_name", name)) { rc = set_error("OBJ without file_name"); break; }
            std::vector<float> verts;
            uint64_t ntris = 0;
            std::cout << "Loading .obj file...\n";
            if ((rc = obj_load(resolve(name, scene_file), verts, ntris)) != 0) break;
            std::cout << "Loaded " << ntris << " triangles\n";
            if (all_verts.size() / 9 + ntris >= 0x40000000ull) { rc = set_error("more than 2^30 triangles"); break; }
            m.first_face = (uint32_t)(all_verts.size() / 9);
            if (sc->meshes.empty()) all_verts.swap(verts);
            else all_verts.insert(all_verts.end(), verts.begin(), verts.end());
            sc->meshes.push_back(m);
        }
    }
    if (rc == 0 && !sc->meshes.empty()) {
        std::cout << "Constructing BVH...\n";
        const int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        rc = bvh_build(all_verts.data(), all_verts.size() / 9, threads, &sc->bvh);
        // the first mesh's material doubles as the scene's single mesh material, which is all a one-OBJ scene needs
        std::memcpy(sc->desc.mesh_material, sc->meshes[0].material, sizeof sc->desc.mesh_material);
        sc->desc.mesh_material_type = sc->meshes[0].material_type;
    }
    if (rc == 0 && !have_camera) rc = set_error("%s: no CAMERA block", scene_file);
    if (rc == 0 && !have_env) rc = set_error("%s: no ENVIRONMENT block", scene_file);
    if (rc != 0) {
        pth_scene_free(sc);
        return rc;
    }
    pt_scene_desc &d = sc->desc;
    d.bvh_nodes = sc->bvh.nodes; d.num_nodes = sc->bvh.num_nodes;
    d.bvh_tris = sc->bvh.tris; d.num_tri_float4 = sc->bvh.num_tri_float4;
    d.bvh_prim_ids = sc->bvh.prim_ids; d.num_tris = sc->bvh.num_tris;
    d.planes = sc->planes.empty() ? nullptr : sc->planes.data(); d.num_planes = sc->planes.size();
    d.spheres = sc->spheres.empty() ? nullptr : sc->spheres.data(); d.num_spheres = sc->spheres.size();
    d.env_texels = sc->env.data();
    d.env_marginal = sc->marginal.data();
    d.env_conditional = sc->conditional.data();
    if (sc->meshes.size() > 1) { d.meshes = sc->meshes.data(); d.num_meshes = sc->meshes.size(); }
    *out = sc;
    return 0;
}

void pth_scene_free(pth_scene *scene)
{
    if (!scene) return;
    pth_bvh_free(&scene->bvh);
    delete scene;
}

const pt_scene_desc *pth_scene_desc(const pth_scene *scene) { return scene ? &scene->desc : nullptr; }

int pth_obj_load(const char *path, float **out_verts, uint64_t *out_ntris)
{
    if (!path || !out_verts || !out_ntris) return set_error("pth_obj_load: bad argument");
    std::vector<float> v;
    uint64_t n = 0;
    int rc = obj_load(path, v, n);
    if (rc != 0) return rc;
    float *p = (float *)std::malloc(std::max<size_t>(1, v.size()) * sizeof(float));
    if (!p) return set_error("pth_obj_load: out of memory");
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(float));
    *out_verts = p;
    *out_ntris = n;
    return 0;
}

void pth_free(void *p) { std::free(p); }

int pth_bvh_build(const float *verts, uint64_t ntris, int threads, pth_bvh *out)
{
    if (!out) return set_error("pth_bvh_build: out is NULL");
    return bvh_build(verts, ntris, threads, out);
}

void pth_bvh_free(pth_bvh *b)
{
    if (!b) return;
    std::free(b->nodes); std::free(b->tris); std::free(b->prim_ids);
    std::memset(b, 0, sizeof *b);
}

int pth_hdr_load(const char *path, float **out_rgba, uint32_t *out_w, uint32_t *out_h)
{
    if (!path || !out_rgba || !out_w || !out_h) return set_error("pth_hdr_load: bad argument");
    std::vector<float> px;
    int rc = hdr_load(path, px, *out_w, *out_h);
    if (rc != 0) return rc;
    float *p = (float *)std::malloc(px.size() * sizeof(float));
    if (!p) return set_error("pth_hdr_load: out of memory");
    std::memcpy(p, px.data(), px.size() * sizeof(float));
    *out_rgba = p;
    return 0;
}

int pth_hdr_write(const char *path, const float *rgb, uint32_t w, uint32_t h, int channels)
{
    if (!path || !rgb) return set_error("pth_hdr_write: bad argument");
    return hdr_write(path, rgb, w, h, channels);
}

int pth_env_build(float *texels, uint32_t w, uint32_t h, uint64_t *marginal, uint64_t *conditional)
{
    return env_build(texels, w, h, marginal, conditional);
}

void pth_camera_make(const float pos[3], const float look_at[3], float fov, float aspect, float aperture,
                     float focus_dist, pt_camera *out)
{
    camera_make(pos, look_at, fov, aspect, aperture, focus_dist, out);
}

void pth_tonemap16(const float *film, uint64_t npixels, float exposure, float gamma, uint16_t *out)
{
    tonemap16(film, npixels, exposure, gamma, out);
}

int pth_png16_write(const char *path, const uint16_t *rgb, uint32_t w, uint32_t h)
{
    if (!path || !rgb) return set_error("pth_png16_write: bad argument");
    return png16_write(path, rgb, w, h);
}

int pth_pfm_write(const char *path, const float *film_rgb, uint32_t w, uint32_t h)
{
    if (!path || !film_rgb || w == 0 || h == 0) return set_error("pth_pfm_write: bad argument");
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return set_error("%s: cannot open for writing", path);
    std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    bool ok = true;
    for (uint32_t row = h; row-- > 0 && ok;)           // the film's row 0 is the top, a PFM starts at the bottom
        ok = std::fwrite(film_rgb + (size_t)row * w * 3, sizeof(float), (size_t)w * 3, f) == (size_t)w * 3;
    ok = std::fclose(f) == 0 && ok;
    return ok ? 0 : set_error("%s: write failed", path);
}

int pth_asset_generate(const char *kind, const char *path)
{
    if (!kind || !path) return set_error("pth_asset_generate: bad argument");
    return asset_generate(kind, path);
}

}   // extern "C"
