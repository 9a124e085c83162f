"""The calibration as one binary (mst_calibration_save / _load; SURVEY section 5's "checkpoint" row): round trip of every
field, rejection of damaged blobs, and an engine calibrated from a reloaded blob producing the same panorama."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
import stitchingvideo_amd as S
from stitchingvideo_amd import capi, rig as rigmod


def _rig_with_everything():
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    w, h = rig["size"]
    K = np.array([[w * 0.52, 0, w * 0.49], [0, w * 0.52, h * 0.51], [0, 0, 1]])
    rig["cameras"][2]["undistort"] = S.init_undistort_rectify_map(K, [-0.3, 0.1, 0.001, -0.002], (w, h))
    rig["cameras"][3]["T"] = (0.1, -0.2, 0.05)
    m = rig["cameras"][4]["mask"]
    rig["cameras"][4]["mask"] = np.ascontiguousarray(np.pad(m, ((0, 0), (0, 7))))[:, :m.shape[1]]      # a padded row stride
    rig["cameras"][5]["seam_mask"] = (np.random.default_rng(1).random((33, 57)) > 0.4).astype(np.uint8) * 255
    return rig


def test_round_trip_and_damage():
    rig = _rig_with_everything()
    blob = S.save_calibration(rig["scale"], rig["cameras"])
    assert isinstance(blob, bytes) and len(blob) % 8 == 0
    scale, cams = S.load_calibration(blob)
    assert scale == np.float32(rig["scale"])
    assert len(cams) == len(rig["cameras"])
    for a, b in zip(rig["cameras"], cams):
        assert tuple(a["size"]) == tuple(b["size"])
        assert np.array_equal(np.asarray(a["K"], np.float32).reshape(3, 3), b["K"])
        assert np.array_equal(np.asarray(a["R"], np.float32).reshape(3, 3), b["R"])
        assert np.array_equal(np.asarray(a.get("T", (0, 0, 0)), np.float32), b["T"])
        assert np.array_equal(np.asarray(a["mask"]), b["mask"])
        assert ("gain_map" in b) == (a.get("gain_map") is not None)
        if "gain_map" in b:
            assert np.array_equal(np.asarray(a["gain_map"], np.float32), b["gain_map"])
        assert ("undistort" in b) == (a.get("undistort") is not None)
        if "undistort" in b:
            assert np.array_equal(a["undistort"][0], b["undistort"][0]) and np.array_equal(a["undistort"][1], b["undistort"][1])
        assert ("seam_mask" in b) == (a.get("seam_mask") is not None)
        if "seam_mask" in b:
            assert np.array_equal(a["seam_mask"], b["seam_mask"])
    assert S.save_calibration(scale, cams) == blob                           # canonical: saving the loaded one gives the same bytes
    # damage: a flipped payload byte, a truncated blob, a wrong magic, too little room for the cameras
    L = capi.load()
    cams_c = (capi.Camera * capi.MAX_CAMERAS)()
    cal = capi.Calibration()

    def load(b, room=capi.MAX_CAMERAS):
        raw = np.frombuffer(b, np.uint8).copy()
        return L.mst_calibration_load(raw.ctypes.data, raw.size, C.byref(cal), cams_c, room)

    assert load(blob) == 0
    bad = bytearray(blob); bad[len(bad) // 2] ^= 0x40
    assert load(bytes(bad)) == capi.ERR_INVALID_ARG
    assert load(blob[:len(blob) - 8]) == capi.ERR_INVALID_ARG
    bad = bytearray(blob); bad[0] ^= 1
    assert load(bytes(bad)) == capi.ERR_INVALID_ARG
    assert load(blob, room=3) == capi.ERR_INVALID_ARG
    assert L.mst_calibration_blob_size(None) == 0
    assert L.mst_calibration_save(None, None, 0, None) == capi.ERR_INVALID_ARG


@pytest.mark.gpu
def test_engine_from_a_reloaded_blob():
    rig = rigmod.build_rig(rigmod.scaled_config("C3", 8))
    frames = rigmod.frames_for(rig, 0)
    eng = H.make_engine(rig)
    pano, mask = eng.compose(frames)
    scale, cams = S.load_calibration(S.save_calibration(rig["scale"], rig["cameras"]))
    eng2 = S.PanoramaEngine(len(cams), warp=rig["warp"], blend=rig["blend"], expos=rig["expos"])
    eng2.update(scale, cams)
    pano2, mask2 = eng2.compose(frames)
    assert np.array_equal(pano, pano2) and np.array_equal(mask, mask2)
    eng.close(); eng2.close()


def _fnv1a(b):
    h = 0xcbf29ce484222325
    for x in b:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def test_loader_survives_crafted_blobs():
    """Random damage WITH the checksum repaired (so that the field validators are what stands between the file and the
    pointers): the loader must answer OK or INVALID_ARG, and every pointer it hands out must lie inside the blob."""
    import struct
    rig = rigmod.build_rig(rigmod.scaled_config("C2", 16))
    blob = bytearray(S.save_calibration(rig["scale"], rig["cameras"]))
    L = capi.load()
    rng = np.random.default_rng(5)
    header = 32                      # magic, version, cameras, scale, total_bytes, checksum
    records_end = header + 2 * 168   # two camera records (see csrc/calib_blob.cpp)
    accepted = 0
    for trial in range(400):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(8, records_end))                 # header fields after the magic, and the records
            if header - 8 <= pos < header:
                continue                                            # (the checksum itself is rewritten below)
            b[pos] = int(rng.integers(0, 256))
        if trial % 5 == 0:                                          # a huge offset that would wrap a naive bound check
            struct.pack_into("<Q", b, records_end - 8 * int(rng.integers(1, 6)), 0xFFFFFFFFFFFFFFF8)
        struct.pack_into("<Q", b, 24, _fnv1a(bytes(b[header:])))
        raw = np.frombuffer(bytes(b), np.uint8).copy()
        cams = (capi.Camera * capi.MAX_CAMERAS)()
        cal = capi.Calibration()
        rc = L.mst_calibration_load(raw.ctypes.data, raw.size, C.byref(cal), cams, capi.MAX_CAMERAS)
        assert rc in (0, capi.ERR_INVALID_ARG)
        if rc == 0:
            accepted += 1
            lo, hi = raw.ctypes.data, raw.ctypes.data + raw.size
            for i in range(cal.num_cameras):
                c = cams[i]
                for ptr, n in ((c.gain_map, 4 * c.gain_w * c.gain_h), (c.mask, c.mask_width * c.mask_height),
                               (c.undistort_map1, 4 * c.src_width * c.src_height), (c.undistort_map2, 2 * c.src_width * c.src_height),
                               (c.seam_mask, c.seam_mask_width * c.seam_mask_height)):
                    if ptr:
                        assert lo <= ptr and ptr + n <= hi
    assert accepted > 0            # (some damage only touches values, not structure: K, R, gains)
