"""The C++ host side (pathtracer_amd/host): image codecs, OBJ/MTL loading, and -- on the GPU -- the `pt`
command-line driver end to end (JSON config -> Scene / Camera / Renderer -> PNG), compared with the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

from pathtracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "pathtracer_amd", "libpt_host.so")
PT = os.path.join(ROOT, "pathtracer_amd", "pt")


@pytest.fixture(scope="module")
def host():
    assert os.path.exists(HOST_LIB), "build the host library first (__graft_entry__.build())"
    L = C.CDLL(HOST_LIB)
    L.pt_host_image_load.restype = C.c_void_p
    L.pt_host_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pt_host_free.argtypes = [C.c_void_p]
    L.pt_host_image_write.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    L.pt_host_load_obj.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p] * 7 + [C.POINTER(C.c_int)]
    return L


def _load(host, path):
    w, h = C.c_int(), C.c_int()
    p = host.pt_host_image_load(str(path).encode(), C.byref(w), C.byref(h))
    if not p:
        return None
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, 3)).copy()
    host.pt_host_free(p)
    return a


def test_png_decode_variants(host, tmp_path):
    rng = np.random.RandomState(1)
    rgb = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    cases = {
        "rgb": Image.fromarray(rgb),
        "rgba": Image.fromarray(np.dstack([rgb, rng.randint(0, 256, (37, 53)).astype(np.uint8)]), "RGBA"),
        "grey": Image.fromarray(rgb[..., 0], "L"),
        "palette": Image.fromarray(rgb).quantize(64),
        "grey16": Image.fromarray((rgb[..., 0].astype(np.uint16) * 257), "I;16"),
        "bilevel": Image.fromarray(rgb[..., 0] > 127),
    }
    for name, im in cases.items():
        p = tmp_path / f"{name}.png"
        im.save(p)
        got = _load(host, p)
        assert got is not None, name
        expect = np.asarray(Image.open(p).convert("RGB"))
        if name == "grey16":  # 16-bit samples keep their high byte (Pillow's conversion saturates instead)
            expect = np.repeat(rgb[..., :1], 3, axis=2)
        # rows come back bottom-up, as the reference's loader flips them (src/image.cpp:29)
        assert np.array_equal(got, np.flipud(expect)), name
    assert _load(host, tmp_path / "missing.png") is None
    (tmp_path / "junk.png").write_bytes(b"not a png at all")
    assert _load(host, tmp_path / "junk.png") is None
    inter = tmp_path / "interlaced.png"
    # Pillow cannot write Adam7; flip the header's interlace byte (and leave the CRC: it is not checked)
    raw = bytearray((tmp_path / "rgb.png").read_bytes())
    raw[28] = 1
    inter.write_bytes(bytes(raw))
    assert _load(host, inter) is None


def test_png_encode_round_trip(host, tmp_path):
    rng = np.random.RandomState(2)
    rgb = rng.randint(0, 256, (29, 41, 3)).astype(np.uint8)
    p = tmp_path / "out.png"
    assert host.pt_host_image_write(str(p).encode(), rgb.ctypes.data, 41, 29) == 1
    assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), rgb)  # row 0 of the buffer = top of the file


def test_jpeg_baseline_decode(host, tmp_path):
    ys, xs = np.mgrid[0:96, 0:160]
    smooth = np.stack([128 + 100 * np.sin(xs / 23.0), 128 + 90 * np.cos(ys / 17.0), 60 + xs * 0.8], -1)
    img = Image.fromarray(np.clip(smooth, 0, 255).astype(np.uint8))
    for name, kw in {"444": dict(subsampling=0), "420": dict(subsampling=2), "422": dict(subsampling=1),
                     "restart": dict(subsampling=2, restart_marker_blocks=3)}.items():
        p = tmp_path / f"{name}.jpg"
        try:
            img.save(p, quality=92, **kw)
        except TypeError:
            continue
        got = _load(host, p)
        assert got is not None, name
        expect = np.flipud(np.asarray(Image.open(p).convert("RGB"))).astype(int)
        d = np.abs(got.astype(int) - expect)
        assert d.mean() < 1.5 and d.max() <= 12, (name, d.mean(), d.max())  # IDCT / chroma upsampling latitude
    grey = tmp_path / "grey.jpg"
    img.convert("L").save(grey, quality=90)
    got = _load(host, grey)
    assert got is not None and np.abs(got[..., 0].astype(int) - np.flipud(np.asarray(Image.open(grey))).astype(int)).max() <= 2
    prog = tmp_path / "prog.jpg"
    img.save(prog, quality=90, progressive=True)
    assert _load(host, prog) is None  # reported as unsupported, not mis-decoded


def test_obj_loader_matches_the_procedural_cornell_box(host, tmp_path):
    """Quads are split like the reference's OBJ reader, numbers pass through float, material attributes follow
    the illum / Kd / Ke mapping of src/scene.cpp:127-152, and emissive triangles become lights."""
    cfg = scenes.write_cornell_assets(str(tmp_path), 32, 32)
    sd = scenes.cornell_box(32, 32)
    n = 64
    pos, nrm, uv = np.zeros((n, 9)), np.zeros((n, 9)), np.zeros((n, 6))
    mtype, tex = np.zeros(n, np.int32), np.zeros(n, np.int32)
    alb, emi = np.zeros((n, 3)), np.zeros((n, 3))
    nl = C.c_int()
    k = host.pt_host_load_obj(os.path.join(str(tmp_path), "cornell_box.obj").encode(), n, pos.ctypes.data, nrm.ctypes.data,
                              uv.ctypes.data, mtype.ctypes.data, alb.ctypes.data, emi.ctypes.data, tex.ctypes.data, C.byref(nl))
    assert k == 12 and nl.value == 2
    assert np.array_equal(pos[:12].reshape(12, 3, 3), sd.tri_pos)
    assert np.array_equal(nrm[:12].reshape(12, 3, 3), sd.tri_nrm)
    assert np.array_equal(uv[:12].reshape(12, 3, 2), sd.tri_uv)
    for i in range(12):
        m = sd.materials[sd.tri_mat[i]]
        assert mtype[i] == m.type and np.allclose(alb[i], m.albedo) and np.allclose(emi[i], m.emission)
        assert tex[i] == (1 if m.texture >= 0 else 0)
    assert os.path.exists(cfg)


def test_obj_loader_polygons_and_relative_indices(host, tmp_path):
    (tmp_path / "p.obj").write_text(
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\nvn 0 0 1\n"
        "f -5//1 -4//1 -3//1 -2//1 -1//1\n"   # pentagon, relative indices -> fan of 3 triangles
        "f 1 2 3\n"                            # no normals, no uvs
        "v 3 0 0\nv 2 0 0\nv 2 0.1 0\nv 3 5 0\n"
        "f 6 7 8 9\n")                         # diagonal 6-8 shorter than 7-9 -> [0,1,2],[0,2,3]
    n = 16
    pos, nrm, uv = np.zeros((n, 9)), np.zeros((n, 9)), np.zeros((n, 6))
    mtype, tex = np.zeros(n, np.int32), np.zeros(n, np.int32)
    alb, emi = np.zeros((n, 3)), np.zeros((n, 3))
    k = host.pt_host_load_obj(str(tmp_path / "p.obj").encode(), n, pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data,
                              mtype.ctypes.data, alb.ctypes.data, emi.ctypes.data, tex.ctypes.data, None)
    assert k == 6
    P = pos[:6].reshape(6, 3, 3)
    assert np.allclose(P[0], [[0, 0, 0], [1, 0, 0], [1, 1, 0]]) and np.allclose(P[2], [[0, 0, 0], [0, 1, 0], [0.5, 1.5, 0]])
    assert np.allclose(nrm[0].reshape(3, 3), [[0, 0, 1]] * 3) and not nrm[3].any()
    assert np.allclose(P[4], [[3, 0, 0], [2, 0, 0], [2, 0.1, 0]]) and np.allclose(P[5], [[3, 0, 0], [2, 0.1, 0], [3, 5, 0]])
    assert (mtype[:6] == scenes.DIFFUSE).all() and np.allclose(alb[:6], 0.5)  # no mtllib: the default material


@pytest.mark.gpu
@pytest.mark.parametrize("with_spheres", [False, True])
def test_pt_command_line_matches_oracle(tmp_path, with_spheres):
    """pt <config> <out> <samples> <bounces> <batch>: argv order, JSON keys and the batch loop of
    src/main.cpp:172-270; the PNG must match the oracle's tone-mapped render of the same scene and seed."""
    from oracle import orc
    cfg = scenes.write_cornell_assets(str(tmp_path), 64, 64, spheres=with_spheres)
    out = tmp_path / "result.png"
    env = dict(os.environ, PT_SEED="5")
    r = subprocess.run([PT, cfg, str(out), "32", "4", "16"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "32/32 samples per pixel" in r.stdout and "Render time:" in r.stdout and "Primitive Count: %d" % (14 if with_spheres else 12) in r.stdout
    assert (tmp_path / "latest.png").exists()
    got = np.asarray(Image.open(out).convert("RGB")).astype(int)
    sd = scenes.cornell_box(64, 64)
    if with_spheres:
        scenes.add_json_spheres(sd)
    o = orc.OracleScene(sd)
    ref, _ = o.render(32, 4, seed=5)
    ref8 = orc.tonemap_u8(ref).astype(int)
    d = np.abs(got - ref8)
    print("mean abs diff", d.mean())
    assert d.mean() < 2.0
    # usage message and a missing config are errors, as in the reference
    assert subprocess.run([PT], capture_output=True).returncode == 1
    assert subprocess.run([PT, str(tmp_path / "nope.json"), str(out)], capture_output=True).returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("devices,split", [("0", "spp"), ("0,0", "tile"), ("0,0,0", "spp")])
def test_pt_with_environment_map_file_and_textured_sphere(host, tmp_path, devices, split):
    """The asset path end to end at the size of the reference's own environment map: a 3584 x 1792 baseline JPEG
    (the reference's prater.jpg itself does not travel with the tests; this one is generated) decoded by the host
    library, 19 MB of texels uploaded, sampled by the kernels as background; plus a sphere with a PNG texture
    (JSON key spheres[].texture, src/main.cpp:75,135-142).  Also with the GPU named two and three times in
    PT_DEVICES, so that the host API's multi-GPU path (ptk_group) runs on a one-GPU machine.  The oracle renders
    the same scene from the texels the host decoder produced."""
    from oracle import orc
    env = scenes.procedural_envmap(3584, 1792)
    Image.fromarray(np.flipud(env)).save(tmp_path / "sky.jpg", quality=90, subsampling=2)
    globe = scenes.procedural_texture(256, 6, seed=3)
    Image.fromarray(np.flipud(globe)).save(tmp_path / "globe.png")
    cfg = scenes.write_cornell_assets(str(tmp_path), 80, 48, background_texture=str(tmp_path / "sky.jpg"),
                                      textured_sphere=str(tmp_path / "globe.png"))
    out = tmp_path / "result.png"
    r = subprocess.run([PT, cfg, str(out), "24", "3", "8", "--seed", "9", "--split", split], capture_output=True, text=True,
                       env=dict(os.environ, PT_DEVICES=devices), timeout=180)
    assert r.returncode == 0, r.stderr
    assert "Loaded environment texture" in r.stdout and "Loaded texture" in r.stdout and "24/24 samples per pixel" in r.stdout
    got = np.asarray(Image.open(out).convert("RGB")).astype(int)
    decoded = _load(host, tmp_path / "sky.jpg")  # bottom-up rows, as the kernels and the oracle take them
    assert decoded.shape == (1792, 3584, 3) and np.abs(decoded.astype(int) - env.astype(int)).mean() < 3.0
    sd = scenes.cornell_box(80, 48)
    tex = sd.add_texture(globe)
    sd.add_sphere((0.5, 1.2, 1.5), 1.2, sd.add_material(scenes.Material(albedo=(1.0, 1.0, 1.0), texture=tex)))
    sd.background_texture = sd.add_texture(decoded)
    ref, counters = orc.OracleScene(sd).render(24, 3, seed=9)
    d = np.abs(got - orc.tonemap_u8(ref).astype(int))
    print(devices, split, "mean abs diff", d.mean())
    assert d.mean() < 2.0
    # print_stats (src/geometry.cpp:273-278) reports whole-run totals, however many devices shared the frame
    import re
    printed = {k: int(v) for k, v in re.findall(r"(Path Segments|Intersection Queries): (\d+)", r.stdout)}
    assert abs(printed["Path Segments"] - counters["segments"]) <= 0.005 * counters["segments"], (printed, counters)
    assert abs(printed["Intersection Queries"] - counters["queries"]) <= 0.005 * counters["queries"], (printed, counters)


def test_tree_cache_round_trip(tmp_path, monkeypatch):
    """PTK_BVH_CACHE: the second build of the same geometry reads the stored tree (same nodes, same hits); a
    damaged file is rejected by the validity check and rebuilt; other geometry gets its own file."""
    import numpy as np
    from tests import raygen
    from tests.backends import HostBackend, scene
    from pathtracer_amd import scenes

    monkeypatch.setenv("PTK_BVH_CACHE", str(tmp_path))
    sd = scene("mesh")
    rays = np.asarray(raygen.interior_rays(5000, 9), np.float64).astype(np.float32).astype(np.float64)
    first = HostBackend(sd)
    assert first.info()["tree_from_cache"] == 0
    files = list(tmp_path.glob("ptkbvh-*.bin"))
    assert len(files) == 1
    hits1, _ = first.intersect(rays)
    second = HostBackend(sd)
    assert second.info()["tree_from_cache"] == 1
    assert second.info()["n_nodes"] == first.info()["n_nodes"] and second.info()["max_depth"] == first.info()["max_depth"]
    hits2, _ = second.intersect(rays)
    assert (hits1["id"] == hits2["id"]).all() and (hits1["t"] == hits2["t"]).all()

    # corrupt one child link in the middle of the node array: the file must be refused, the tree rebuilt
    raw = bytearray(files[0].read_bytes())
    header = 8 + 16 + 16
    link = header + 32 * 10 + 24  # child0 of node 10
    raw[link:link + 4] = (0x7fffff00).to_bytes(4, "little")
    files[0].write_bytes(bytes(raw))
    third = HostBackend(sd)
    assert third.info()["tree_from_cache"] == 0
    hits3, _ = third.intersect(rays)
    assert (hits1["id"] == hits3["id"]).all()
    # a truncated file likewise
    files[0].write_bytes(files[0].read_bytes()[:1000])
    assert HostBackend(sd).info()["tree_from_cache"] == 0

    other = scenes.cornell_with_mesh(3, 32, 32)
    assert HostBackend(other).info()["tree_from_cache"] == 0
    assert len(list(tmp_path.glob("ptkbvh-*.bin"))) == 2
    assert not list(tmp_path.glob("*.tmp*"))


@pytest.mark.parametrize("scene_name", ["cornell", "mesh", "glass", "sky"])
def test_grid_boxes_contain_their_primitives(scene_name):
    """Node boxes are stored as 16-bit grid positions rounded outwards: every primitive must lie inside the
    intersection of all boxes on its path (checked in double precision), be reached exactly once, and no leaf
    may lie deeper than the traversal stack allows."""
    from tests.backends import HostBackend, scene
    b = HostBackend(scene(scene_name))
    r = b.k.tree_check()
    info = b.info()
    assert r["violations"] == 0, r
    assert r["reached"] == info["n_prims"], (r, info)
    assert r["deepest"] == info["max_depth"] <= 32, (r, info)


def test_obj_loader_pieces_give_the_single_thread_result(host, tmp_path, monkeypatch):
    """Files above 4 MB are parsed in pieces by several threads.  Relative indices, usemtl switches and polygons
    next to the piece borders must come out exactly as from one thread; an out-of-range index is reported with
    its line number whichever piece holds it."""
    rng = np.random.RandomState(1)
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 0 0\nnewmtl b\nKd 0 1 0\nKe 1 1 1\n")
    lines, nv = ["mtllib m.mtl"], 0
    for blk in range(40000):
        k = int(rng.randint(3, 7))
        for _ in range(k):
            lines.append("v %.5f %.5f %.5f" % tuple(rng.uniform(-1, 1, 3)))
            lines.append("vt %.4f %.4f" % tuple(rng.uniform(0, 1, 2)))
        nv += k
        if blk % 7 == 0:
            lines.append("usemtl " + ("a" if (blk // 7) % 2 else "b"))
        if blk % 3 == 0:
            lines.append("f " + " ".join("%d/%d" % (-k + j, -k + j) for j in range(k)))
        else:
            lines.append("f " + " ".join("%d/%d" % (nv - k + j + 1, nv - k + j + 1) for j in range(k)))
    text = "\n".join(lines) + "\n"
    assert len(text) > (4 << 20)
    (tmp_path / "t.obj").write_text(text)

    def load(threads):
        monkeypatch.setenv("PT_LOAD_THREADS", str(threads))
        n = 110000
        a = [np.zeros((n, 9)), np.zeros((n, 9)), np.zeros((n, 6)), np.zeros(n, np.int32), np.zeros((n, 3)), np.zeros((n, 3)),
             np.zeros(n, np.int32)]
        nl = C.c_int()
        r = host.pt_host_load_obj(str(tmp_path / "t.obj").encode(), n, *[x.ctypes.data for x in a], C.byref(nl))
        return r, nl.value, [x[:r].copy() for x in a]

    one = load(1)
    assert one[0] > 90000 and one[1] > 0
    for threads in (5, 16):
        many = load(threads)
        assert many[0] == one[0] and many[1] == one[1]
        for x, y in zip(many[2], one[2]):
            assert np.array_equal(x, y)

    # a bad reference deep in the file: the process exits like the reference's loader, naming the line
    bad_line = len(lines) - 1000
    lines.insert(bad_line, "f 1 2 99999999")
    (tmp_path / "bad.obj").write_text("\n".join(lines) + "\n")
    code = ("import ctypes as C; L = C.CDLL(%r); L.pt_host_load_obj.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p] * 8; "
            "L.pt_host_load_obj(%r, 0, *[None] * 8)" % (HOST_LIB, str(tmp_path / "bad.obj").encode()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, PT_LOAD_THREADS="8"))
    assert r.returncode == 1 and ("at line %d" % (bad_line + 1)) in r.stderr, r.stderr[-300:]


def test_threaded_tree_builder_under_address_sanitizer(tmp_path):
    """The host builder hands node slots to its threads from one array sized up front.  With 16 or more
    threads (the default on the GPU box) uniform random triangles used to push the nodes built above the
    per-thread blocks past the end of that array; the builder is compiled here with the address sanitizer and
    run at the thread counts and sizes that overflowed."""
    exe = tmp_path / "build_stress"
    src = [os.path.join(ROOT, "tests", "hostcheck", "build_stress.cpp"), os.path.join(ROOT, "pathtracer_amd", "csrc", "ptk_build.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-pthread", "-o", str(exe)] + src)
    for n, threads, crowd in ((140000, 16, 0), (200000, 24, 0), (300000, 32, 0), (200000, 64, 1)):
        r = subprocess.run([str(exe), str(n), "3", str(crowd)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PTK_BUILD_THREADS=str(threads)))
        assert r.returncode == 0 and r.stdout.startswith("ok"), (n, threads, r.stdout[-200:], r.stderr[-600:])


def test_image_decoders_reject_damaged_files_under_address_sanitizer(tmp_path):
    """Asset files are untrusted input: truncated segments, Huffman table selectors above 3, quantisation
    tables cut short and PNG bit depths that the format does not define must be refused, never read out of
    bounds.  The decoders are compiled with the address and undefined-behaviour sanitizers and fed damaged files."""
    exe = tmp_path / "codec_fuzz"
    host_dir = os.path.join(ROOT, "pathtracer_amd", "host")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-o", str(exe), os.path.join(ROOT, "tests", "hostcheck", "codec_fuzz.cpp"),
                           os.path.join(host_dir, "codec_jpeg.cpp"), os.path.join(host_dir, "codec_png.cpp"), "-lz"])
    rng = np.random.RandomState(7)
    ys, xs = np.mgrid[0:24, 0:40]
    img = Image.fromarray(np.clip(np.stack([128 + 100 * np.sin(xs / 5.0), 128 + 90 * np.cos(ys / 4.0), 60 + xs * 4.0], -1), 0, 255).astype(np.uint8))
    jpg, png, pal = tmp_path / "a.jpg", tmp_path / "a.png", tmp_path / "p.png"
    img.save(jpg, quality=85, subsampling=2)
    img.save(png)
    img.quantize(16).save(pal, bits=4)
    files = []

    def emit(name, data):
        p = tmp_path / name
        p.write_bytes(bytes(data))
        files.append(str(p))

    for src in (jpg, png, pal):
        raw = src.read_bytes()
        emit("whole_" + src.name, raw)
        for cut in sorted(set(list(range(0, min(len(raw), 700), 3)) + [len(raw) - k for k in range(1, 40)])):
            emit(f"cut{cut}_{src.name}", raw[:cut])
        for k in range(300):
            b = bytearray(raw)
            for _ in range(int(rng.randint(1, 4))):
                b[int(rng.randint(2, len(b)))] = int(rng.randint(0, 256))
            emit(f"mut{k}_{src.name}", b)
    raw = bytearray(jpg.read_bytes())
    sos = raw.find(b"\xff\xda")
    assert sos > 0
    for sel in (0x40, 0x0f, 0xff):  # table selectors index four-entry arrays
        b = bytearray(raw)
        b[sos + 6] = sel
        emit(f"sel{sel}.jpg", b)
    for depth in (3, 5, 6, 7):  # not PNG bit depths: would make the sample shift negative
        b = bytearray(pal.read_bytes())
        b[24] = depth
        emit(f"depth{depth}.png", b)
    for i in range(0, len(files), 400):
        r = subprocess.run([str(exe)] + files[i:i + 400], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-300:], r.stderr[-1500:])
    first = subprocess.run([str(exe), files[0], str(tmp_path / "sel64.jpg"), str(tmp_path / "depth3.png")], capture_output=True, text=True)
    assert "decoded 1 rejected 2" in first.stdout, first.stdout


REFERENCE_ASSETS = "/root/reference/assets"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ASSETS), reason="the reference checkout is not on this machine")
def test_the_references_own_asset_files(host):
    """The files the reference ships for its one scene, read as data by this build's readers: the configuration by
    the driver's JSON reader (against Python's), the Cornell box OBJ + MTL by the OBJ reader (it must give the
    procedural scene the benchmarks use, bit for bit), the two textures by the PNG and JPEG decoders (against
    Pillow's).  Runs where the reference checkout exists; the GPU box has none and needs none."""
    import json
    # configuration
    cfg = os.path.join(REFERENCE_ASSETS, "cornell_box", "cornell_box.json")
    host.pt_host_json_flatten.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    need = host.pt_host_json_flatten(cfg.encode(), None, 0)
    assert need > 0
    buf = C.create_string_buffer(need)
    assert host.pt_host_json_flatten(cfg.encode(), buf, need) == need
    got = dict(line.split("=", 1) for line in buf.value.decode().splitlines())

    def flatten(v, at, out):
        if isinstance(v, dict):
            for k, x in v.items():
                flatten(x, f"{at}.{k}" if at else k, out)
        elif isinstance(v, list):
            for i, x in enumerate(v):
                flatten(x, f"{at}[{i}]", out)
        elif isinstance(v, bool):
            out[at] = "true" if v else "false"
        elif isinstance(v, (int, float)):
            out[at] = "%.17g" % float(v)
        else:
            out[at] = str(v)
        return out
    want = flatten(json.load(open(cfg)), "", {})
    assert got == want and got["camera_fov"] == "14.5" and got["spheres[2].hidden"] == "true"

    # geometry and materials
    sd = scenes.cornell_box(32, 32)
    n = 64
    pos, nrm, uv = np.zeros((n, 9)), np.zeros((n, 9)), np.zeros((n, 6))
    mtype, tex = np.zeros(n, np.int32), np.zeros(n, np.int32)
    alb, emi = np.zeros((n, 3)), np.zeros((n, 3))
    nl = C.c_int()
    k = host.pt_host_load_obj(os.path.join(REFERENCE_ASSETS, "cornell_box", "cornell_box.obj").encode(), n, pos.ctypes.data,
                              nrm.ctypes.data, uv.ctypes.data, mtype.ctypes.data, alb.ctypes.data, emi.ctypes.data,
                              tex.ctypes.data, C.byref(nl))
    assert k == 12 and nl.value == 2
    assert np.array_equal(pos[:12].reshape(12, 3, 3), sd.tri_pos)
    assert np.array_equal(nrm[:12].reshape(12, 3, 3), sd.tri_nrm)
    assert np.array_equal(uv[:12].reshape(12, 3, 2), sd.tri_uv)
    for i in range(12):
        m = sd.materials[sd.tri_mat[i]]
        assert mtype[i] == m.type and np.allclose(alb[i], m.albedo) and np.allclose(emi[i], m.emission)
        assert tex[i] == (1 if m.texture >= 0 else 0)

    # textures
    png = os.path.join(REFERENCE_ASSETS, "textures", "uv-test.png")
    got_png = _load(host, png)
    assert got_png is not None and np.array_equal(got_png, np.flipud(np.asarray(Image.open(png).convert("RGB"))))
    jpg = os.path.join(REFERENCE_ASSETS, "textures", "prater.jpg")
    got_jpg = _load(host, jpg)
    expect = np.flipud(np.asarray(Image.open(jpg).convert("RGB"))).astype(int)
    assert got_jpg is not None and got_jpg.shape == expect.shape == (1792, 3584, 3)
    d = np.abs(got_jpg.astype(int) - expect)
    assert d.mean() < 1.5 and np.percentile(d, 99.9) <= 12  # IDCT / chroma upsampling latitude, as for the generated files
