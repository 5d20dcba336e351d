"""CPU: the C-ABI library loads and exports every symbol include/cgrt.h declares; host-side prerequisites
(mesh loaders, bump mesh, tree build, lens stream) of the PRODUCT match the golden vectors.  No compute calls."""
import ctypes as C
import hashlib
import json
import os
import re
import sys

import numpy as np
import pytest

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)
import make_golden  # noqa: E402

META = json.load(open(os.path.join(GOLD, "trace_meta.json")))


def test_library_exports_every_declared_symbol():
    from cgraytracing_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "cgrt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cgrt_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 17
    lib = C.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libcgrt.so does not export %s" % name
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    assert _capi.lib().cgrt_version() == 112


def test_struct_layouts_match_header():
    from cgraytracing_amd import _capi
    assert C.sizeof(_capi.Camera) == 48
    assert C.sizeof(_capi.Grid) == 56  # 12 x int32 + uint64
    assert C.sizeof(_capi.SceneStats) == 8 * 4 + 4 * 8
    assert C.sizeof(_capi.Photons) == 88       # 6 doubles, int64, 2 x int32, uint64, double, int64 (pinned in tests/native/abi_c99.c too)
    assert C.sizeof(_capi.PpmResult) == 96     # 3 pointers, 5 x uint64, 4 doubles


def test_error_reporting_without_gpu_or_bad_args():
    import cgraytracing_amd as cg
    from cgraytracing_amd import _capi
    L = _capi.lib()
    assert L.cgrt_scene_create(None) == -1 and b"null" in L.cgrt_last_error()
    with pytest.raises(ValueError):
        cg.Bezier([(0, 0, 1)] * 7, (0, 0, 0), (1, 1, 1))
    s = cg.Scene(scenes.scene_c1(), commit=False)
    # tracing an uncommitted scene is refused, never a crash or a silent CPU path
    with pytest.raises(_capi.CgrtError) as e:
        s.trace_grid_host(8, 8)
    assert e.value.code == -1
    # a plane that names an unknown texture id
    h = s._h
    d3 = (C.c_double * 3)(0, 1, 0)
    assert L.cgrt_scene_add_plane(h, d3, d3, d3, 0.0, 0.0, 5) == -1
    s.close()
    # `vector<Object*> objs` has no bound (main.cpp:277): neither has the scene (round 2 stopped at 96, the LDS list's size)
    big = cg.Scene([scenes.Sphere((0, 0, 30 + i), 1, (1, 1, 1)) for i in range(96)], commit=False)
    assert L.cgrt_scene_add_sphere(big._h, d3, 1.0, d3, 0.0, 0.0) == 96
    assert big.stats()["n_objects"] == 97
    big.close()


def test_malformed_mesh_is_io_error(tmp_path):
    import cgraytracing_amd as cg
    from cgraytracing_amd import _capi
    p = tmp_path / "bad.txt"
    p.write_text("begin\nvertex 0 0 0\nvertex 1 0 0\nend\n")
    with pytest.raises(_capi.CgrtError) as e:
        cg.Scene([cg.TriangleMesh(str(p), 1, (0, 0, 0), (1, 1, 1))], commit=False)
    assert e.value.code == -2
    empty = cg.Scene([cg.TriangleMesh(str(tmp_path / "missing.txt"), 1, (0, 0, 0), (1, 1, 1))], commit=False)
    assert empty.stats()["n_triangles"] == 0
    empty.close()


@pytest.mark.parametrize("name", ["t0", "t1", "t2"])
def test_product_loaders_and_tree_build_match_reference(name):
    import cgraytracing_amd as cg
    file, a, b, typ = make_golden.LOADER_CASES[name]
    g = np.load(os.path.join(GOLD, "loader_%s.npz" % name))
    m = cg.TriangleMesh(os.path.join(GOLD, "assets", file), a, b, (0.6, 0.7, 0.9), 0.8, 0.5, typ)
    s = cg.Scene(scenes.planes() + [m], commit=False)
    nodes, leaf, bbox, tris = s.tree_dump(0)
    assert np.array_equal(tris, g["tris"])
    assert np.array_equal(nodes, g["nodes"]) and np.array_equal(leaf, g["leaf"]) and np.array_equal(bbox, g["bbox"])
    s.close()


@pytest.mark.parametrize("case,key", [("bunny_glass_chess_64", "mesh_tree"), ("dragon_64", "mesh_tree"),
                                      ("stone_bump_64x48", "bump_tree"), ("chess_bump_48x36", "bump_tree"),
                                      ("stone_full_bump_64x48", "bump_tree"), ("glass_bump_floor_48x36", "bump_tree")])
def test_product_tree_fingerprints(case, key):
    import cgraytracing_amd as cg
    mk = {c[0]: c[1] for c in make_golden.trace_cases()}[case]
    s = cg.Scene(mk(), commit=False)
    nodes, leaf, bbox, tris = s.tree_dump(0)
    assert make_golden.fingerprint(nodes, leaf) == META[case][key]
    if key == "bump_tree":
        assert hashlib.sha256(tris.tobytes()).hexdigest() == META[case]["bump_tris_sha256"]
    st = s.stats()
    assert st["n_trees"] == 1 and st["n_nodes"] == len(nodes) and st["n_triangles"] == len(leaf)
    s.close()


def test_product_lens_stream_matches_reference():
    from cgraytracing_amd import _capi
    g = np.load(os.path.join(GOLD, "lens_samples.npz"))
    pix = np.ascontiguousarray(g["pix"], np.int64)
    smp = np.ascontiguousarray(g["smp"], np.int32)
    out = np.zeros((len(pix), 3))
    _capi.check(_capi.lib().cgrt_lens_samples(int(g["seed"]), pix.ctypes.data, smp.ctypes.data, len(pix), 1.5,
                                              out.ctypes.data))
    assert np.array_equal(out, g["out"])


def test_product_never_touches_the_oracle():
    """The shipped path must not import, link or fall back to anything under oracle/."""
    pk = os.path.join(ROOT, "cgraytracing_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hpp", ".hip")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liborc" not in txt and "oracle/" not in txt.replace("oracle/).", ""), os.path.join(dp, f)
                assert "backends" not in txt


def test_header_is_plain_c99(tmp_path):
    """include/cgrt.h compiles as strict C99 and the C caller links against libcgrt.so."""
    import subprocess
    from cgraytracing_amd import _capi
    src = os.path.join(ROOT, "tests", "native", "abi_c99.c")
    obj = str(tmp_path / "abi.o")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", src, "-o", obj])
    so = str(tmp_path / "libabi.so")
    subprocess.check_call(["gcc", "-shared", "-o", so, obj, "-L", os.path.dirname(_capi.LIB_PATH), "-lcgrt",
                           "-Wl,-rpath," + os.path.dirname(_capi.LIB_PATH)])
    assert C.CDLL(so).cgrt_abi_smoke() == 0


def test_host_builder_under_sanitizers(tmp_path):
    """Loaders, bump-mesh construction and tree build (cgrt_build.cpp) run clean under ASan + UBSan (CPU build;
    GPU sanitizers are not available on the pool)."""
    import subprocess
    exe = str(tmp_path / "build_san")
    csrc = os.path.join(ROOT, "cgraytracing_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", csrc, os.path.join(ROOT, "tests", "native", "build_san.cpp"),
                           os.path.join(csrc, "cgrt_build.cpp"), "-o", exe])
    bad = tmp_path / "bad.txt"
    bad.write_text("3\nv 0 0 0\nv 1 0 0\n")
    assets = os.path.join(GOLD, "assets")
    out = subprocess.run([exe, os.path.join(assets, "mesh_t0.txt"), "0", os.path.join(assets, "mesh_t1.txt"), "1",
                          os.path.join(assets, "mesh_t2.txt"), "2", str(bad), "1", str(tmp_path / "missing.txt"), "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ERROR" not in out.stderr and "runtime error" not in out.stderr
    assert "bad.txt -> -2" in out.stdout  # malformed file refused
    assert "missing.txt -> " in out.stdout and "objs=" in out.stdout


def test_threaded_host_build_under_tsan(tmp_path):
    """The multi-threaded parts of the host build (reference-order tree with preassigned node numbers, SAH halves, the
    eight octant layouts) on a 44 402-triangle mesh under ThreadSanitizer, and -- same program, plain build -- twice in
    a row with identical output (the reference-order arrays must not depend on thread timing)."""
    import subprocess
    csrc = os.path.join(ROOT, "cgraytracing_amd", "csrc")
    exe = str(tmp_path / "build_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", csrc,
                           os.path.join(ROOT, "tests", "native", "build_san.cpp"), os.path.join(csrc, "cgrt_build.cpp"),
                           "-o", exe])
    out = subprocess.run([exe, "--big"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-2000:]
    assert "big: tris=44402" in out.stdout


def _png_decode(path):
    """Minimal PNG reader (8-bit RGB, filter 0 only -- what cgrt_write_png emits) with full CRC / Adler checks via zlib."""
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == bytes([0x89]) + b"PNG" + bytes([0x0d, 0x0a, 0x1a, 0x0a])
    pos, chunks = 8, []
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + data) & 0xffffffff == crc, typ
        chunks.append((typ, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, flt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)  # checks adler32
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (300, 211), (768, 1024)])
def test_write_png_round_trip(tmp_path, shape):
    """cgrt_write_png (stbi_write_png's role at main.cpp:412) is host-only: a strict decode returns the same bytes;
    PIL, where present, agrees."""
    from cgraytracing_amd.engine import write_png
    h, w = shape
    img = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    path = str(tmp_path / "t.png")
    write_png(path, img)
    assert np.array_equal(_png_decode(path), img)
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), img)


def test_write_png_errors(tmp_path):
    from cgraytracing_amd import _capi
    L = _capi.lib()
    buf = np.zeros(12, np.uint8)
    assert L.cgrt_write_png(None, 2, 2, buf.ctypes.data) == -1
    assert L.cgrt_write_png(os.fsencode(str(tmp_path / "no_such_dir" / "x.png")), 2, 2, buf.ctypes.data) == -2
    assert b"cannot open" in L.cgrt_last_error()


def _check_hierarchy(sc, t):
    """Structural properties the device traversal relies on, per octant copy: preorder with
    consistent skip links; every leaf-order triangle below exactly one leaf; each node's box contains its children's
    boxes and every triangle below it; the two children of a node are ordered near-to-far for the octant on some axis.
    Leaf-level form (transparent owner): the leaves are exactly the reference's leaves.  Triangle-level form (opaque
    owner): groups of <= 4 triangles of the hierarchy's own order, a permutation of the leaf order."""
    nodes, leaf_ids, bbox, tris = sc.tree_dump(t)
    tri_level, order = sc.bvh_order(t)
    box, skip, leaf = sc.bvh_dump(t)
    n = box.shape[1]
    tri = tris[leaf_ids]  # leaf order
    if tri_level:
        assert sorted(order.tolist()) == list(range(len(tri)))
        tri = tri[order]  # the hierarchy's own order
    else:
        ref_leaves = sorted((int(bbox_i), int(nodes[i, 2])) for i, bbox_i in enumerate(range(len(nodes))) if nodes[i, 0] < 0 and nodes[i, 2] > 0)
        assert n == 2 * len(ref_leaves) - 1
    tlo = tri.reshape(-1, 3, 3).min(1)
    thi = tri.reshape(-1, 3, 3).max(1)
    want_leaves = None
    for o in range(8):
        sk, lf, bx = skip[o], leaf[o], box[o]
        assert sk[0] == n
        got = sorted((int(v) >> 4, int(v) & 15) for v in lf[lf >= 0])
        if want_leaves is None:
            want_leaves = got
            assert sum(c for _, c in got) == len(tri)
            assert [f for f, _ in got] == list(np.cumsum([0] + [c for _, c in got[:-1]]))  # contiguous runs
            assert max(c for _, c in got) <= (4 if tri_level else 9)
            if not tri_level:  # exactly the reference's leaves
                assert got == sorted((int(leaf_first), int(cnt)) for leaf_first, cnt in _ref_leaf_runs(nodes))
        assert got == want_leaves
        sgn = np.array([-1.0 if (o >> k) & 1 else 1.0 for k in range(3)])
        for i in range(n):  # node i's subtree is [i, skip[i])
            assert i < sk[i] <= n
            if lf[i] >= 0:
                assert sk[i] == i + 1
                f, c = int(lf[i]) >> 4, int(lf[i]) & 15
                assert (bx[i, :3] <= tlo[f:f + c].min(0)).all() and (bx[i, 3:] >= thi[f:f + c].max(0)).all()
            else:
                a, b = i + 1, sk[i + 1]
                assert sk[b] == sk[i]  # exactly two children
                for ch in (a, b):
                    assert (bx[ch, :3] >= bx[i, :3]).all() and (bx[ch, 3:] <= bx[i, 3:]).all()
                ca, cb = bx[a, :3] + bx[a, 3:], bx[b, :3] + bx[b, 3:]
                assert ((cb - ca) * sgn >= 0).any()


def _check_wide(sc, t):
    """The 4-wide form the device walks for an opaque owner: a tree over node 0 in which every node is referenced exactly
    once; its leaves are exactly the leaves of the one-box-per-node form, each once, with that form's (grown, outward-rounded)
    box; an inner reference carries the box of the node it stands for = the union of that node's child boxes; the stack bound
    holds (children - 1 summed along any root-to-leaf path)."""
    box, ref, need = sc.wide_dump(t)
    tri_level, _ = sc.bvh_order(t)
    if not tri_level:
        assert len(ref) == 0
        return
    bbox, bskip, bleaf = sc.bvh_dump(t)
    want = {int(v): bbox[0][i] for i, v in enumerate(bleaf[0]) if v >= 0}
    NONE = -2 ** 31
    seen_nodes, seen_leaves = set(), {}
    worst = 0
    stack = [(0, 0)]
    while stack:
        i, pending = stack.pop()
        assert i not in seen_nodes
        seen_nodes.add(i)
        kids = [k for k in range(4) if ref[i, k] != NONE]
        assert kids == list(range(len(kids))) and len(kids) >= 1
        for k in kids:
            r = int(ref[i, k])
            if r >= 0:
                assert r not in seen_leaves
                seen_leaves[r] = box[i, k]
                worst = max(worst, pending + len(kids) - 1)
            else:
                c = ~r
                ck = [q for q in range(4) if ref[c, q] != NONE]
                assert (box[c, ck, :3].min(0) == box[i, k, :3]).all() and (box[c, ck, 3:].max(0) == box[i, k, 3:]).all()
                stack.append((c, pending + len(kids) - 1))
    assert seen_nodes == set(range(len(ref)))
    assert set(seen_leaves) == set(want)
    for r, b in seen_leaves.items():
        assert (b == want[r]).all()
    assert worst == need < 64


def _ref_leaf_runs(nodes):
    """(first index in leaf order, count) of the reference's non-empty leaves, in node order."""
    runs, first = [], 0
    for i in range(len(nodes)):
        if nodes[i, 0] < 0:
            if nodes[i, 2] > 0:
                runs.append((first, int(nodes[i, 2])))
            first += int(nodes[i, 2])
    return runs


def test_device_hierarchy_structure():
    """Host build of the SAH hierarchies (no GPU): a small authored mesh as glass and as an opaque object, the glass
    bunny, the opaque dragon-sized procedural mesh, a transparent bump floor (an opaque one is walked as a grid)."""
    from cgraytracing_amd.engine import Scene
    path = os.path.join(GOLD, "assets", "mesh_t1.txt")
    for transp in (0.5, 0.0):
        sc = Scene([scenes.TriangleMesh(path, 2.5, (-3.0, -6.0, 28.0), (0.6, 0.7, 0.9), 0.8, transp, 1)], commit=False)
        assert sc.bvh_order(0)[0] == (transp == 0.0)
        _check_hierarchy(sc, 0)
        _check_wide(sc, 0)
        sc.close()
    sc = Scene(scenes.scene_c3(True), commit=False)
    _check_hierarchy(sc, 0)
    sc.close()
    sc = Scene([scenes.TriangleMesh.from_triangles(scenes.procedural_mesh(40, 30, (-5.0, -10.0, 30.0), 9.0), (0.25, 0.25, 0.5), 0.0, 0.0, 1)],
               commit=False)
    _check_hierarchy(sc, 0)
    _check_wide(sc, 0)
    sc.close()
    # geometrically growing triangles: binned SAH peels the largest few off per level -- a chain hundreds of levels deep,
    # beyond the wide walk's stack; the build must fall back to shallower (median) splits and still meet the bound
    k = np.arange(400, dtype=np.float64)
    s = 1e-3 * 1.15 ** k
    chain = np.stack([np.stack([s, 0 * s, 30 + 0 * s], 1), np.stack([2 * s, 0 * s, 30 + 0 * s], 1), np.stack([s, s, 30 + 0 * s], 1)], 1)
    sc = Scene([scenes.TriangleMesh.from_triangles(chain, (0.5, 0.5, 0.5), 0.0, 0.0)], commit=False)
    _check_hierarchy(sc, 0)
    _check_wide(sc, 0)
    sc.close()
    floor = scenes.Plane((0.0, -20, 0), (0, 1, 0), (0.15, 0.15, 0.15), 0.8, 0.5, scenes.stone_small_texture(True))
    sc = Scene([floor], commit=False)
    assert not sc.bvh_order(0)[0]
    _check_hierarchy(sc, 0)
    sc.close()
    sc = Scene(scenes.planes(scenes.stone_small_texture(True)), commit=False)
    assert sc.bvh_dump(0)[0].shape[1] == 0  # opaque bump floor: height-field walk, no hierarchy
    sc.close()


def test_library_load_puts_torch_first():
    """One HIP runtime per process: libtorch_hip asks for "libamdhip64.so" (its bundled copy), libcgrt.so for
    "libamdhip64.so.7"; only with torch loaded FIRST does our NEEDED entry resolve to the copy already in the process.
    build() followed by smoke() in one process (library first, torch later) found "no ROCm-capable device" before
    _capi.lib() imported torch itself.  Checked here without a GPU: loading the library leaves torch imported."""
    import subprocess
    import sys
    code = ("import sys; from cgraytracing_amd import _capi; assert 'torch' not in sys.modules; _capi.lib(); "
            "assert 'torch' in sys.modules; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
