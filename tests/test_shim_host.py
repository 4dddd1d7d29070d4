"""CPU tests of the host-only parts of the C++ shim (tests/cxx/host_driver.cpp):
the reference's on-disk text formats -- transformation.txt
(mvr/src/point_cloud.cpp:305-347) and axis.txt (mvr/src/registrator.cpp:258-308)
--, the OSG<->Eigen matrix bridge (mvr/include/types.h:20-50), the turntable
prior (point_cloud.cpp:400-413) and Registrator::refineAxis (:402-455)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PIV = np.array([-13.382786, 50.223461, 917.4776])
AX = np.array([-0.054323, -0.814921, -0.577020])


@pytest.fixture(scope="module")
def host(built, tmp_path_factory):
    built.build_cxx_tests()
    exe = os.path.join(ROOT, "tests", "cxx", "host_driver")
    d = tmp_path_factory.mktemp("mvrfiles")
    r = subprocess.run([exe, str(d)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout), d


def colvec(m16):
    """row-vector (OSG) 4x4 printed row-major -> column-vector matrix."""
    return np.array(m16).reshape(4, 4).T


def test_turntable_prior(host, orc):
    out, _ = host
    assert np.array_equal(np.array(out["prior_view0"]).reshape(4, 4), np.eye(4))
    pf, af = PIV.astype(np.float32).astype(np.float64), AX.astype(np.float32).astype(np.float64)
    for v, key in ((1, "prior_view1"), (7, "prior_view7")):
        exp = orc.axis_rotation(pf, af, orc.turntable_angle(v, 12))       # registrator.cpp:331-342
        assert np.abs(colvec(out[key]) - exp).max() < 1e-9
    assert out["init_keeps_pose"] == 1          # initRotation only touches identity poses (point_cloud.cpp:402)


def test_transformation_txt_format(host):
    out, d = host
    assert out["save_tf"] == 1 and out["load_tf"] == 1 and out["load_missing"] == 0
    txt = open(os.path.join(d, "transformation.txt")).read()
    rows = txt.strip("\n").split("\n")
    assert len(rows) == 4 and all(len(r.split()) == 4 for r in rows) and all(r.endswith(" ") for r in rows)
    # the file holds the COLUMN-vector matrix row by row (matrix(j,i), i outer), %lf = 6 decimals
    T = colvec(out["prior_view1"])
    file_T = np.array([[float(x) for x in r.split()] for r in rows])
    assert np.abs(file_T - T).max() <= 5.01e-7
    assert all(len(x.split(".")[1]) == 6 for r in rows for x in r.split())
    assert np.abs(colvec(out["loaded_view1"]) - file_T).max() == 0          # what was written is what is read back
    assert np.allclose(file_T[3], [0, 0, 0, 1])


def test_axis_txt_format(host):
    out, d = host
    assert out["save_axis"] == 1 and out["load_axis"] == 1
    lines = open(os.path.join(d, "axis.txt")).read().strip().split("\n")
    assert len(lines) == 2
    got = np.array([[float(x) for x in ln.split()] for ln in lines])
    assert np.abs(got[0] - PIV).max() < 1e-4 and np.abs(got[1] - AX).max() < 1e-6
    assert np.allclose(out["axis_loaded"], np.concatenate([got[0], got[1]]), atol=1e-4)


def test_pcl_matrix_caster_is_a_transpose(host):
    out, _ = host
    osg = np.array(out["prior_view1"]).reshape(4, 4)
    e = np.array(out["caster_e"]).reshape(4, 4)             # Matrix4f printed as (r,c)
    assert np.array_equal(e.astype(np.float32), osg.T.astype(np.float32))
    assert np.array_equal(np.array(out["caster_back"]).reshape(4, 4).astype(np.float32), osg.astype(np.float32))


def test_refine_axis_recovers_the_turntable(host):
    out, _ = host
    piv, ax = np.array(out["refined"][:3]), np.array(out["refined"][3:])
    a = AX / np.linalg.norm(AX)
    assert abs(abs(ax @ a) - 1) < 1e-6                      # axis direction recovered
    off = piv - PIV
    assert np.linalg.norm(off - (off @ a) * a) < 1e-2       # pivot back on the true axis line


def test_pcd_round_trips_and_rejects(host):
    out, d = host
    for mode in ("ascii", "binary", "compressed"):
        assert out["pcd"][mode] == [1, 1, 1], mode                  # saved, loaded, identical (ascii prints %.9g: exact for f32)
    assert out["pcd"]["foreign"][:2] == [1, 3] and out["pcd"]["foreign"][2:] == [12.0, 22.0, 32.0]      # field order z y x, f64, extra u8 field
    assert out["pcd"]["rejects"] == [0, 0, 0, 2]                    # truncated payload, no x y z, missing file; cloud untouched
    assert out["pcd"]["path"] == "/data/ws/points/object_00007/view_03/points.pcd"
    assert out["xyz"][0] == 1037 and out["xyz"][2] == 1.0           # data[3] = 1 (pcl::PointXYZ padding)


def test_pcd_bytes_against_independent_reader(host):
    """The three files, decoded by a reader written separately from the format description (tests/pcd_py.py): same header
    vocabulary PCL uses, same values in every encoding, binary_compressed is field-major behind {u32 csize, u32 usize}."""
    import pcd_py
    out, d = host
    recs = {}
    for mode in ("ascii", "binary", "compressed"):
        hdr, rec = pcd_py.read_pcd(os.path.join(d, "cloud_%s.pcd" % mode))
        assert hdr["FIELDS"] == ["x", "y", "z", "rgb", "normal_x", "normal_y", "normal_z", "curvature"]
        assert hdr["SIZE"] == ["4"] * 8 and hdr["TYPE"] == ["F"] * 8 and hdr["WIDTH"] == ["1037"] and hdr["HEIGHT"] == ["1"]
        assert hdr["VIEWPOINT"] == ["0", "0", "0", "1", "0", "0", "0"] and hdr["VERSION"] == ["0.7"]
        recs[mode] = rec
    for f in ("x", "y", "z", "normal_x", "normal_y", "normal_z", "curvature"):
        assert np.array_equal(recs["binary"][f], recs["compressed"][f])
        assert np.array_equal(recs["binary"][f], recs["ascii"][f])
    rgb = recs["binary"]["rgb"].view(np.uint32)
    assert np.array_equal(rgb, recs["compressed"]["rgb"].view(np.uint32))
    assert np.array_equal((rgb >> 8) & 255, np.arange(1037) & 255) and np.array_equal(rgb & 255, (np.arange(1037) * 7) & 255)
    assert (rgb >> 24).max() == 0
    # the compressed file really is compressed (smooth fields), and smaller than the binary one
    assert os.path.getsize(os.path.join(d, "cloud_compressed.pcd")) < os.path.getsize(os.path.join(d, "cloud_binary.pcd"))
    # points.asc: "%f %f %f %d %d %d"
    lines = open(os.path.join(d, "points.asc")).read().strip().split("\n")
    assert out["pcd"]["asc"] == 1 and len(lines) == 1037
    x, y, z, r, g, b = lines[5].split()
    assert abs(float(x) - recs["binary"]["x"][5]) < 1e-6 and len(x.split(".")[1]) == 6 and int(g) == 5 and int(b) == 35


def test_scan_cloud_open_and_save(host):
    """ScanCloud::open (point_cloud.cpp:78-95): PCD + transformation.txt beside it, registered = non-identity pose;
    ScanCloud::save: "*.ply" = XYZ in the turntable's canonical frame (point_cloud.cpp:99-113: pivot -> origin,
    axis -> +z) as ASCII PLY, anything else = binary_compressed PCD with colours / normals."""
    out, d = host
    assert out["open"] == [0, 1, 3, 3, 1, 9]          # missing file refused; 3 points, rich records, registered, colour kept
    file_T = np.array([[float(x) for x in r.split()] for r in open(os.path.join(d, "transformation.txt")).read().strip("\n").split("\n")])
    assert np.abs(colvec(out["opened_pose"]) - file_T).max() == 0
    assert out["ply"] == [1, 1, 1, 1, 3, 3, 9]
    canon = np.array(out["canon"]).reshape(3, 3)
    assert np.abs(canon[0]).max() < 1e-4                                            # the pivot goes to the origin
    assert np.abs(canon[1] - [0, 0, 10]).max() < 1e-4                               # 10 mm along the axis -> +z
    assert abs(np.linalg.norm(canon[2]) - 5.0) < 1e-4                               # a rigid motion
    head = open(os.path.join(d, "canon.ply")).read().split("end_header")[0].split("\n")
    assert head[0] == "ply" and head[1] == "format ascii 1.0" and "element vertex 3" in head and "element camera 1" in head
    assert [l for l in head if l.startswith("property")][:3] == ["property float x", "property float y", "property float z"]
    # a foreign PLY (double coordinates in z, q, x, y order, an empty face element) and a binary one (refused, cloud untouched)
    assert out["foreign_ply"] == [1, 2, 4.0, 5.0, 6.0, 0, 1]


def test_save_registered_points(host, orc):
    """Registrator::saveRegisteredPoints (registrator.cpp:344-400): registered views only, in view order; points and
    normals both go through the full pose (the reference translates its normals too)."""
    out, d = host
    assert out["merged"] == [9, 1, 9, 41, 0]           # views 0 (4 pts) + 1 (5 pts); view 2 is not registered
    # view 0 has the identity pose: untouched
    assert out["merged_p0"] == [0.0, -3.0, 900.0, 0.0, np.float32(0.6), np.float32(0.8)]
    T = colvec(out["prior_view1"])
    p = np.array([10.0, -3.0, 901.0]); n = np.array([0.0, np.float32(0.6), np.float32(0.8)])
    exp_p, exp_n = T[:3, :3] @ p + T[:3, 3], T[:3, :3] @ n + T[:3, 3]
    got = np.array(out["merged_p4"])
    assert np.abs(got[:3] - exp_p).max() < 1e-4 and np.abs(got[3:] - exp_n).max() < 1e-4
    lines = open(os.path.join(d, "merged", "points.asc")).read().strip().split("\n")
    assert len(lines) == 9 and lines[4].split()[3:] == ["41", "0", "7"]
