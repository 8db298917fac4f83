"""CPU tests of the EXR codec behind bcd::ImageIO (OpenEXR itself is absent on both boxes): round trips through the
library's writer, and files assembled independently in Python from the published format (uncompressed, ZIP, ZIPS, RLE)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import bcd_amd.core as core


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def _pack_block(raw, comp):
    if comp == 0:
        return raw
    a = np.frombuffer(raw, np.uint8)
    half = (a.size + 1) // 2
    t = np.concatenate([a[0::2], a[1::2]])
    assert t[:half].size == half
    d = t.astype(np.int32)
    d[1:] = (t[1:].astype(np.int32) - t[:-1].astype(np.int32) + 128) & 0xFF
    d = d.astype(np.uint8).tobytes()
    if comp in (2, 3):
        z = zlib.compress(d)
        return z if len(z) < len(raw) else raw
    out = bytearray()          # RLE: runs only (count >= 0 -> repeat next byte count+1 times), literal runs for singles
    i = 0
    while i < len(d):
        j = i
        while j + 1 < len(d) and d[j + 1] == d[i] and j - i < 126:
            j += 1
        if j > i:
            out += struct.pack("b", j - i) + d[i:i + 1]
            i = j + 1
        else:
            k = i
            while k + 1 < len(d) and d[k + 1] != d[k] and k - i < 126:
                k += 1
            n = k - i + 1
            out += struct.pack("b", -n) + d[i:i + n]
            i += n
    return bytes(out) if len(out) < len(raw) else raw


def build_exr(path, planes, types, comp, min_xy=(0, 0)):
    """planes: dict name -> HxW array; types: dict name -> 'half' | 'float' | 'uint'"""
    names = sorted(planes)
    H, W = planes[names[0]].shape
    ch = b""
    code = {"uint": 0, "half": 1, "float": 2}
    for n in names:
        ch += n.encode() + b"\0" + struct.pack("<iBBBBii", code[types[n]], 0, 0, 0, 0, 1, 1)
    ch += b"\0"
    x0, y0 = min_xy
    box = struct.pack("<iiii", x0, y0, x0 + W - 1, y0 + H - 1)
    hdr = struct.pack("<II", 20000630, 2)
    hdr += _attr("channels", "chlist", ch) + _attr("compression", "compression", bytes([comp]))
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    lines = 16 if comp == 3 else 1
    nblocks = (H + lines - 1) // lines
    chunks = []
    for b in range(nblocks):
        raw = b""
        for l in range(b * lines, min(H, (b + 1) * lines)):
            for n in names:
                dt = {"half": np.float16, "float": np.float32, "uint": np.uint32}[types[n]]
                raw += planes[n][l].astype(dt).tobytes()
        data = _pack_block(raw, comp)
        chunks.append(struct.pack("<ii", y0 + b * lines, len(data)) + data)
    off = len(hdr) + 8 * nblocks
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    open(path, "wb").write(hdr + table + b"".join(chunks))


@pytest.mark.parametrize("comp", [0, 1, 2, 3])
def test_reads_independently_built_files(tmp_path, comp):
    rng = np.random.default_rng(comp)
    H, W = 37, 23
    planes = {"R": rng.random((H, W)), "G": rng.random((H, W)) * 3, "B": rng.random((H, W)), "A": np.ones((H, W)), "Z": rng.random((H, W))}
    planes["G"][5:20, 3:15] = 0.25  # runs for RLE
    types = {"R": "half", "G": "float", "B": "half", "A": "half", "Z": "float"}
    p = str(tmp_path / "c.exr")
    build_exr(p, planes, types, comp, min_xy=(3, -2))
    rgb = core.read_exr(p, False)
    assert rgb.shape == (H, W, 3)
    assert np.array_equal(rgb[..., 1], planes["G"].astype(np.float32))
    assert np.array_equal(rgb[..., 0], planes["R"].astype(np.float16).astype(np.float32))
    allc = core.read_exr(p, True)                 # alphabetical: A, B, G, R, Z
    assert allc.shape == (H, W, 5)
    assert np.array_equal(allc[..., 4], planes["Z"].astype(np.float32)) and (allc[..., 0] == 1).all()


def test_multichannel_roundtrip_is_exact(tmp_path):
    rng = np.random.default_rng(1)
    img = (rng.standard_normal((41, 29, 61)) * 10).astype(np.float32)
    img[3, 4, 5] = np.float32(1e-30)
    p = str(tmp_path / "h.exr")
    core.write_exr(p, img, True)
    back = core.read_exr(p, True)
    assert back.shape == img.shape and np.array_equal(back.view(np.uint32), img.view(np.uint32))
    raw = open(p, "rb").read()
    assert raw[:4] == struct.pack("<I", 20000630) and b"Bin_0000" in raw and b"Bin_0060" in raw


def test_colour_write_is_half_rgba(tmp_path):
    rng = np.random.default_rng(2)
    img = (rng.random((18, 33, 3)) * 4).astype(np.float32)
    img[0, 0] = [70000.0, 1e-9, 6.1e-5]            # overflow -> inf, underflow -> 0, half subnormal range
    p = str(tmp_path / "c.exr")
    core.write_exr(p, img, False)
    back = core.read_exr(p, False)
    with np.errstate(over="ignore"):
        want = img.astype(np.float16).astype(np.float32)
    assert np.array_equal(back, want)              # round-to-nearest-even like numpy's float16
    allc = core.read_exr(p, True)
    assert allc.shape[-1] == 4 and (allc[..., 0] == 1).all()   # A, B, G, R with A = 1 (io_exr.cpp:147-164 of the reference)
    grey = np.repeat(img[..., :1], 3, -1)
    core.write_exr(p, grey, False)
    assert core.read_exr(p, False).shape[-1] == 1  # R == G == B collapses to depth 1, like the reference's loadEXR


def test_errors_are_reported_not_fatal(tmp_path):
    with pytest.raises(IOError):
        core.read_exr(str(tmp_path / "missing.exr"), True)
    p = str(tmp_path / "bad.exr")
    open(p, "wb").write(b"not an exr file at all")
    with pytest.raises(IOError):
        core.read_exr(p, False)
    planes = {"R": np.zeros((4, 4))}
    build_exr(p, planes, {"R": "half"}, 0)
    raw = bytearray(open(p, "rb").read())
    raw[raw.index(b"compression\0compression\0") + 28] = 4   # PIZ
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError, match="compression"):
        core.read_exr(p, False)


def test_cli_usage_and_argument_errors():
    exe = os.path.join(os.path.dirname(core.LIB_PATH), "bcd_cli")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Missing required program argument(s): -i -h -c -o" in r.stdout
    for flag in ("-o", "-i", "-h", "-c", "-d", "-b", "-w", "-r", "-p", "--p-factor", "-m", "-s", "--ncores", "--use-cuda", "-e"):
        assert flag in r.stdout
    r = subprocess.run([exe, "-m", "1.5", "-o", "x.exr"], capture_output=True, text=True)
    assert r.returncode == 1 and "between 0 and 1" in r.stdout
    r = subprocess.run([exe, "-i", "/nonexistent.exr"], capture_output=True, text=True)
    assert r.returncode == 1 and "couldn't load input color image" in r.stdout


def test_bcd_json_presets_roundtrip(tmp_path):
    import ctypes as C
    import json
    lib = core.lib()
    src = tmp_path / "scene.bcd.json"
    src.write_text(json.dumps({"inputColorFile": "img/frame.exr", "inputHistoFile": "img/frame_hist.exr", "inputCovarFile": "img/frame_cov.exr",
                               "nbOfScales": 2, "histoDistanceThreshold": 0.75, "searchWindowRadius": 9, "randomPixelOrder": False,
                               "markedPixelsSkippingProbability": 0.5, "minEigenValue": 1e-6, "useCuda": True, "nbOfCores": 4,
                               "performSpikeRemovalPrefiltering": False, "spikeRemovalThresholdStDevFactor": 2.5,
                               "someFutureKey": {"nested": [1, 2, {"x": "}"}]}}, indent=2))
    out = tmp_path / "copy.bcd.json"
    S, tau, b, r, m, e, sp, sf = C.c_int(), C.c_float(), C.c_int(), C.c_int(), C.c_float(), C.c_float(), C.c_int(), C.c_float()
    path = C.create_string_buffer(512)
    rc = lib.bcdcore_presets_roundtrip(str(src).encode(), str(out).encode(), C.byref(S), C.byref(tau), C.byref(b), C.byref(r), C.byref(m),
                                       C.byref(e), C.byref(sp), C.byref(sf), path, 512)
    assert rc == 0
    assert (S.value, b.value, r.value, sp.value) == (2, 9, 0, 0)
    assert abs(tau.value - 0.75) < 1e-7 and abs(m.value - 0.5) < 1e-7 and abs(e.value - 1e-6) < 1e-12 and abs(sf.value - 2.5) < 1e-6
    assert path.value.decode() == str(tmp_path) + "/img/frame.exr"       # relative to the folder of the .bcd.json file
    back = json.loads(out.read_text())                                    # what we write is plain JSON with the reference's keys
    assert back["nbOfScales"] == 2 and back["randomPixelOrder"] is False and back["inputColorFile"] == "img/frame.exr"
    assert set(back) == {"inputColorFile", "inputHistoFile", "inputCovarFile", "performSpikeRemovalPrefiltering",
                         "spikeRemovalThresholdStDevFactor", "nbOfScales", "histoDistanceThreshold", "useCuda", "nbOfCores", "patchRadius",
                         "searchWindowRadius", "randomPixelOrder", "markedPixelsSkippingProbability", "minEigenValue"}
    bad = tmp_path / "bad.bcd.json"
    bad.write_text("{ \"nbOfScales\": }")
    assert lib.bcdcore_presets_roundtrip(str(bad).encode(), None, C.byref(S), C.byref(tau), C.byref(b), C.byref(r), C.byref(m), C.byref(e),
                                         C.byref(sp), C.byref(sf), path, 512) == -1
