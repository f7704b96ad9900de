"""the reference's `.sdm` file format (Map::write / Map::read, src/sdm/map.cpp:489-575) and the export images
(src/sdm/export.cpp:46-95) of the product's host module lama::sdm (include/lama/sdm_io.h), checked against the oracle's
restatement in both directions: oracle writes -> product reads, product writes -> oracle reads, identical pixels."""
import os
import struct

import numpy as np
import pytest

import _oracle as O
import iris_lama_amd.ffi as F
from _cmp import DM_FIELDS, OCC_FIELDS, assert_maps_equal


def _typed(patches, dtype):
    return {k: (np.ascontiguousarray(c).view(dtype).reshape(-1), m) for k, (c, m) in patches.items()}


@pytest.fixture(scope="module")
def built_maps():
    """A particle's maps after a few corridor scans (oracle)."""
    pts, odom, truth = F.corridor_log(4, 360)
    pf = O.PF(O.default_options(particles=1, seed=5))
    pf.set_prior(O.se2(*odom[0]))
    for k in range(5):
        pf.update(pts[k], O.se2(*odom[k]), float(k))
    return pf, pf.dm(0), pf.occ(0)


def test_header_layout_matches_reference_struct(tmp_path, built_maps):
    _, dm, occ = built_maps
    f = str(tmp_path / "dm.sdm")
    F.sdm_write(f, dm.dump(), F.MAP_DISTANCE, 0.05, dm.max_sqdist())
    raw = open(f, "rb").read()
    magic, version, cell_size, patch_length, num_patches, resolution, is3d = struct.unpack_from("<IHxxIIQfB", raw, 0)
    assert (magic, version, cell_size, patch_length, is3d) == (0x6d64732e, 0x0103, 10, 32, 0)     # map.h:72,75,95-103
    assert raw[:4] == b".sdm" and num_patches == len(dm.dump()) and abs(resolution - 0.05) < 1e-8
    assert struct.unpack_from("<I", raw, 32)[0] == dm.max_sqdist()                                  # writeParameters
    assert len(raw) == 32 + 4 + num_patches * (8 + 10240 + 128)
    f2 = str(tmp_path / "occ.sdm")
    F.sdm_write(f2, occ.dump(), F.MAP_OCCUPANCY)
    assert os.path.getsize(f2) == 32 + len(occ.dump()) * (8 + 4096 + 128)


def test_oracle_writes_product_reads(tmp_path, built_maps):
    _, dm, occ = built_maps
    f = str(tmp_path / "a.sdm")
    dm.write(f)
    kind, res, msq, patches = F.sdm_read(f)
    assert kind == F.MAP_DISTANCE and msq == dm.max_sqdist() and abs(res - 0.05) < 1e-8
    assert_maps_equal(_typed(patches, O.DIST_T), dm.dump(), DM_FIELDS, "dm read")
    occ.write(f)
    kind, res, msq, patches = F.sdm_read(f)
    assert kind == F.MAP_OCCUPANCY
    assert_maps_equal(_typed(patches, O.FREQ_T), occ.dump(), OCC_FIELDS, "occ read")


def test_product_writes_oracle_reads(tmp_path, built_maps):
    _, dm, occ = built_maps
    f = str(tmp_path / "b.sdm")
    F.sdm_write(f, dm.dump(), F.MAP_DISTANCE, 0.05, dm.max_sqdist())
    dm2 = O.DM.new(0.05, 32, 0.1)                  # different max distance: must come back from the file
    assert dm2.read(f)
    assert dm2.max_sqdist() == dm.max_sqdist()
    assert_maps_equal(dm2.dump(), dm.dump(), DM_FIELDS, "dm roundtrip")
    F.sdm_write(f, occ.dump(), F.MAP_OCCUPANCY)
    occ2 = O.Occ.new()
    assert occ2.read(f)
    assert_maps_equal(occ2.dump(), occ.dump(), OCC_FIELDS, "occ roundtrip")
    assert not O.DM.new().read(f)                   # cell size mismatch is rejected (map.cpp:545-547)


def test_export_images_match_oracle(tmp_path, built_maps):
    _, dm, occ = built_maps
    im = F.sdm_image(dm.dump(), F.MAP_DISTANCE, 0.05, dm.max_sqdist())
    assert im.shape == dm.image().shape and np.array_equal(im, dm.image())
    im2 = F.sdm_image(occ.dump(), F.MAP_OCCUPANCY)
    ref = occ.image()
    assert np.array_equal(im2, ref)
    assert set(np.unique(ref)) <= {0, 90, 127, 255} and (ref == 255).sum() > 1000 and (ref == 0).sum() > 50
    f = str(tmp_path / "occ.png")
    F.sdm_export_png(f, occ.dump(), F.MAP_OCCUPANCY)
    raw = open(f, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    w, h = struct.unpack_from(">II", raw, 16)
    assert (h, w) == ref.shape
    # decode the stored-deflate stream back and compare the pixels
    import zlib
    pos, idat = 8, b""
    while pos < len(raw):
        n, typ = struct.unpack_from(">I4s", raw, pos)
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack_from(">I", raw, pos + 8 + n)[0] == zlib.crc32(typ + body)
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w + 1)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:], ref)


@pytest.mark.gpu
def test_device_built_map_round_trips_through_sdm(tmp_path):
    """A map built on the GPU, written as .sdm, read back by the oracle's Map::read: identical to the oracle's own map."""
    pts, odom, truth = F.corridor_log(3, 1080)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=1, seed=5))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=1))
    ctx.init(pts[0], pose0)
    f = str(tmp_path / "gpu_dm.sdm")
    F.sdm_write(f, ctx.download_map(0, F.MAP_DISTANCE), F.MAP_DISTANCE, 0.05, pf.dm(0).max_sqdist())
    dm2 = O.DM.new()
    assert dm2.read(f)
    assert_maps_equal(dm2.dump(), pf.dm(0).dump(), DM_FIELDS, "gpu dm via sdm")
    im = F.sdm_image(ctx.download_map(0, F.MAP_OCCUPANCY), F.MAP_OCCUPANCY)
    assert np.array_equal(im, pf.occ(0).image())
