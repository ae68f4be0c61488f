#!/usr/bin/env python3
"""Model compiler: URDF -> collapsed rigid-body tree constants for the HIP simulator.

Reads the robot description that the reference loads through
`gym.load_asset(..., collapse_fixed_joints=True)` (reference
humanoid/envs/base/legged_robot.py:596-615, humanoid/envs/custom/hector_config.py:28-40)
and emits
  * isaac_amd/assets/hector_model.json   (data consumed by the numpy oracle and host code)
  * isaac_amd/csrc/hx_model_data.h       (the same numbers as C constants for the kernels)

This is a build-time tool.  It runs only where the reference checkout is present; its two outputs
are committed, so nothing at run time (tests on the GPU box, bench.py, smoke) reads /root/reference.

What "collapse" means here (PhysX does the same on import): every link reached through a `fixed`
joint is merged into its nearest movable ancestor: masses add, centres of mass combine, inertia
tensors are rotated into the ancestor frame and shifted with the parallel-axis theorem.

Collision geometry: the reference's collision meshes for trunk / hips / thighs are absent from the
checkout (.MISSING_LARGE_BLOBS); they fall back to the primitive boxes of const.xacro:17-19,128-133.
The toe meshes foot_L2.stl / foot_R2.stl are present; their axis-aligned bounding box is used
(PhysX would use the convex hull; for a flat ground both touch at the sole corners).
"""
import json
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("HX_REFERENCE_ROOT", "/root/reference")
URDF = os.path.join(REF, "resources/robots/hector_v2/xacro/robot.urdf")
MESH_DIR = os.path.join(REF, "resources/robots/hector_v2/meshes")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def rpy_to_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n
    return np.array(v)


def parse_origin(el):
    if el is None:
        return np.zeros(3), np.eye(3)
    xyz = vec(el.get("xyz", "0 0 0"))
    R = rpy_to_mat(vec(el.get("rpy", "0 0 0")))
    return xyz, R


def stl_bbox(path):
    """Axis-aligned bounding box of a binary or ASCII STL."""
    with open(path, "rb") as f:
        data = f.read()
    pts = []
    ntri = struct.unpack_from("<I", data, 80)[0] if len(data) >= 84 else 0
    if len(data) == 84 + 50 * ntri:
        for i in range(ntri):
            off = 84 + 50 * i + 12
            pts.append(struct.unpack_from("<9f", data, off))
        pts = np.array(pts).reshape(-1, 3)
    else:
        for line in data.decode("ascii", "ignore").splitlines():
            t = line.split()
            if len(t) == 4 and t[0] == "vertex":
                pts.append([float(t[1]), float(t[2]), float(t[3])])
        pts = np.array(pts)
    return pts.min(0), pts.max(0)


def collapse(urdf_path, sort_children=False):
    """URDF -> list of collapsed bodies (dicts) in depth-first order.  sort_children: visit the movable children of a
    link in alphabetical order of their joint names -- the order Isaac Gym gives bodies and DoFs (the hector-with-arms
    task indexes L leg 0-4, L arm 5-8, R leg 9-13, R arm 14-17: hector_w_arm_env.py:371-373, although the URDF lists
    both legs first); robot.urdf is already in that order."""
    root = ET.parse(urdf_path).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = root.findall("joint")
    children = {}
    for j in joints:
        children.setdefault(j.find("parent").get("link"), []).append(j)
    child_names = {j.find("child").get("link") for j in joints}
    root_link = [n for n in links if n not in child_names]
    assert len(root_link) == 1
    root_link = root_link[0]

    bodies = []  # collapsed bodies, URDF depth-first order

    def link_inertial(name):
        el = links[name].find("inertial")
        if el is None:
            return 0.0, np.zeros(3), np.zeros((3, 3))
        m = float(el.find("mass").get("value"))
        c, R = parse_origin(el.find("origin"))
        i = el.find("inertia")
        I = np.array([[float(i.get("ixx")), float(i.get("ixy")), float(i.get("ixz"))],
                      [float(i.get("ixy")), float(i.get("iyy")), float(i.get("iyz"))],
                      [float(i.get("ixz")), float(i.get("iyz")), float(i.get("izz"))]])
        return m, c, R @ I @ R.T

    def collect(name, p, R, acc):
        """Accumulate link `name` (pose p,R in the collapsed body's frame) and its fixed subtree."""
        m, c, I = link_inertial(name)
        if m > 0:
            acc.append((m, p + R @ c, R @ I @ R.T))
        movable = []
        for j in children.get(name, []):
            jp, jR = parse_origin(j.find("origin"))
            cp, cR = p + R @ jp, R @ jR
            if j.get("type") == "fixed":
                movable += collect(j.find("child").get("link"), cp, cR, acc)
            else:
                movable.append((j, cp, cR))
        return movable

    def build(name, parent_idx, joint, jpos, jrot):
        acc = []
        # A joint about -x/-y/-z becomes one about +x/+y/+z by giving the body a frame turned half a turn about a
        # perpendicular axis (XBot-L.urdf: the two leg_pitch joints); everything of the body is then collected in that frame.
        fix, k = np.eye(3), -1
        if joint is not None:
            axis = vec(joint.find("axis").get("xyz"))
            k = int(np.argmax(np.abs(axis)))
            assert np.allclose(np.abs(axis), np.eye(3)[k]), "joint axes must be along +-x / +-y / +-z of the joint frame"
            if axis[k] < 0:
                fix = -np.eye(3)
                fix[(k + 1) % 3, (k + 1) % 3] = 1.0           # half turn about the next axis: flips k and the third one
        movable = collect(name, np.zeros(3), fix, acc)
        m = sum(a[0] for a in acc)
        com = sum(a[0] * a[1] for a in acc) / m
        I = np.zeros((3, 3))
        for mi, ci, Ii in acc:
            d = ci - com
            I += Ii + mi * (d @ d * np.eye(3) - np.outer(d, d))
        body = {"name": name, "parent": parent_idx, "mass": m, "com": com.tolist(),
                "inertia_com": I.tolist()}
        if joint is not None:
            lim = joint.find("limit")
            body.update({"joint": joint.get("name"), "offset": jpos.tolist(), "axis": k,
                         "lower": float(lim.get("lower")), "upper": float(lim.get("upper")),
                         "velocity": float(lim.get("velocity")), "effort": float(lim.get("effort"))})
            rot = jrot @ fix                                  # child -> parent rotation at q = 0
            if not np.allclose(rot, np.eye(3)):               # both hector assets: identity, key absent (files unchanged)
                body["rot"] = rot.tolist()
        idx = len(bodies)
        bodies.append(body)
        if sort_children:
            movable = sorted(movable, key=lambda m: m[0].get("name"))
        for j, cp, cR in movable:
            build(j.find("child").get("link"), idx, j, cp, cR)

    build(root_link, -1, None, None, None)
    return bodies


def box_corners(center, size):
    c, h = np.array(center), np.array(size) / 2
    return [(c + h * np.array([sx, sy, sz])).tolist() for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]


def leg_and_base_contacts(name_to_idx):
    """The collision primitives of the biped proper (same for robot.urdf and robot_w_arm.urdf): see the module docstring."""
    contacts = [{"body": name_to_idx["base"], "points": box_corners((0, 0, 0), (0.125, 0.19, 0.248)),
                 "source": "const.xacro:17-19 trunk box (body.stl absent)"}]
    for side, sgn in (("L", 1.0), ("R", -1.0)):
        contacts.append({"body": name_to_idx[f"{side}_thigh"],
                         "points": box_corners((0, sgn * 0.0175, -0.09), (0.06, 0.035, 0.18)),
                         "source": "const.xacro:128-133 thigh box (thigh_combined_*2.stl absent)"})
        lo, hi = stl_bbox(os.path.join(MESH_DIR, f"foot_{side}2.stl"))
        ctr, size = (lo + hi) / 2, hi - lo
        contacts.append({"body": name_to_idx[f"{side}_toe"], "points": box_corners(ctr, size),
                         "source": f"foot_{side}2.stl axis-aligned bounding box", "bbox": [lo.tolist(), hi.tolist()]})
    return contacts


def main_full():
    """hector with arms (robot_w_arm.urdf, task hector_full): model data for the oracle only -- no kernel tables yet."""
    urdf = os.path.join(REF, "resources/robots/hector_v2/xacro/robot_w_arm.urdf")
    bodies = collapse(urdf, sort_children=True)
    assert len(bodies) == 19, len(bodies)
    names = [b["name"] for b in bodies]
    assert [b.get("joint") for b in bodies[1:]] == [
        "L_hip_joint", "L_hip_roll_joint", "L_thigh_joint", "L_calf_joint", "L_toe_joint",
        "L_shoulder_yaw_joint", "L_shoulder_pitch_joint", "L_shoulder_roll_joint", "L_elbow_joint",
        "R_hip_joint", "R_hip_roll_joint", "R_thigh_joint", "R_calf_joint", "R_toe_joint",
        "R_shoulder_yaw_joint", "R_shoulder_pitch_joint", "R_shoulder_roll_joint", "R_elbow_joint"], names
    name_to_idx = {b["name"]: i for i, b in enumerate(bodies)}
    contacts = leg_and_base_contacts(name_to_idx)
    # arm links: bounding box of the collision mesh the URDF names, where the file exists (UpperArmRoll_*.stl does not:
    # the roll links get no shape; they matter only for terminate_after_contacts_on, hector_w_arm_config.py:35)
    root = ET.parse(urdf).getroot()
    for link in root.findall("link"):
        nm = link.get("name")
        if nm not in name_to_idx or nm.split("_")[-1] not in ("twist", "shoulder", "roll", "elbow"):
            continue
        col = link.find("collision")
        mesh = col.find("geometry").find("mesh") if col is not None and col.find("geometry") is not None else None
        if mesh is None:
            continue
        path = os.path.normpath(os.path.join(os.path.dirname(urdf), mesh.get("filename")))
        if not os.path.exists(path):
            continue
        o, R = parse_origin(col.find("origin"))
        assert np.allclose(R, np.eye(3))
        lo, hi = stl_bbox(path)
        contacts.append({"body": name_to_idx[nm], "points": box_corners(o + (lo + hi) / 2, hi - lo),
                         "source": os.path.basename(path) + " axis-aligned bounding box", "bbox": [lo.tolist(), hi.tolist()]})
    model = {"source": "resources/robots/hector_v2/xacro/robot_w_arm.urdf (collapse_fixed_joints, children in alphabetical order)",
             "total_mass": sum(b["mass"] for b in bodies), "bodies": bodies, "contacts": contacts}
    with open(os.path.join(ROOT, "isaac_amd/assets/hector_full_model.json"), "w") as f:
        json.dump(model, f, indent=1)
    # ---- C header for the kernels: per-side table (leg 5 + arm 4 bodies of 16 floats, then 5 corner blocks of 24
    #      floats: thigh, toe, twist, shoulder, elbow) and the base constants
    def flit(v):
        t = "%.9g" % v
        if "." not in t and "e" not in t and "n" not in t:
            t += ".0"
        return t + "f"

    def Io(bd):
        c = np.array(bd["com"]); m = bd["mass"]
        I = np.array(bd["inertia_com"]) + m * (c @ c * np.eye(3) - np.outer(c, c))
        return [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    by_body = {c["body"]: c for c in contacts}
    sidec = []
    for side in range(2):
        row = []
        for k in range(9):
            bd = bodies[1 + side * 9 + k]
            row += list(bd["offset"]) + list(bd["mass"] * np.array(bd["com"])) + Io(bd) + [bd["mass"], bd["lower"], bd["upper"], bd["velocity"]]
        for nm in ("thigh", "toe", "twist", "shoulder", "elbow"):
            c = by_body.get(name_to_idx[("L_" if side == 0 else "R_") + nm])
            assert c is not None, nm
            row += [x for pt in c["points"] for x in pt]
        assert len(row) == 9 * 16 + 5 * 24
        sidec += row
    L = ["// GENERATED by tools/compile_urdf.py --full from the reference's robot_w_arm.urdf -- do not edit.",
         "// 19 collapsed bodies / 18 revolute joints in Isaac Gym order: L leg, L arm, R leg, R arm.",
         "#pragma once", "#define HXF_SIDE_STRIDE 264",
         "__device__ static const float HXF_SIDEC[528] = {%s};" % ", ".join(flit(v) for v in sidec),
         "__device__ static const float HXF_BASE_PTS[24] = {%s};" % ", ".join(flit(x) for pt in contacts[0]["points"] for x in pt),
         "static constexpr float HXF_IO[6] = {%s};" % ", ".join(flit(v) for v in Io(bodies[0])),
         "static constexpr float HXF_H[3] = {%s};" % ", ".join(flit(bodies[0]["mass"] * x) for x in bodies[0]["com"]),
         "static constexpr float HXF_MASS0 = %s;" % flit(bodies[0]["mass"]),
         "static constexpr float HXF_EFFORT[18] = {%s};" % ", ".join(flit(b["effort"]) for b in bodies[1:])]
    with open(os.path.join(ROOT, "isaac_amd/csrc/hx_model_data_full.h"), "w") as f:
        f.write("\n".join(L) + "\n")
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.5f" % b["mass"], b.get("joint"), b.get("axis"))
    print("total mass %.5f" % model["total_mass"])


def main_xbot():
    """XBot-L (task humanoid_ppo): model data for the oracle only.  Twelve revolute joints about the local z
    of frames rotated against their parents ("rot"), upper body and hands collapsed into the base."""
    urdf = os.path.join(REF, "resources/robots/XBot/urdf/XBot-L.urdf")
    bodies = collapse(urdf)
    assert len(bodies) == 13, len(bodies)
    assert [b.get("joint") for b in bodies[1:]] == [f"{s}_{j}_joint" for s in ("left", "right") for j in
                                                    ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll")]
    name_to_idx = {b["name"]: i for i, b in enumerate(bodies)}
    contacts = [{"body": 0, "points": box_corners((0, 0, 0.1), (0.4, 0.4, 0.4)), "source": "XBot-L.urdf base_link collision box"}]
    for side in ("left", "right"):
        nm = f"{side}_ankle_roll_link"
        lo, hi = stl_bbox(os.path.join(os.path.dirname(urdf), "../meshes", nm + ".STL"))
        b = bodies[name_to_idx[nm]]
        # the mesh is given in the URDF link frame; a body whose frame was turned (negative axis) would need the same turn
        # here -- the ankle_roll joints are about +z
        assert vec(ET.parse(urdf).getroot().find(f"joint[@name='{side}_ankle_roll_joint']").find("axis").get("xyz"))[2] > 0
        contacts.append({"body": name_to_idx[nm], "points": box_corners((lo + hi) / 2, hi - lo),
                         "source": nm + ".STL axis-aligned bounding box", "bbox": [lo.tolist(), hi.tolist()]})
    model = {"source": "resources/robots/XBot/urdf/XBot-L.urdf (collapse_fixed_joints)",
             "total_mass": sum(b["mass"] for b in bodies), "bodies": bodies, "contacts": contacts}
    with open(os.path.join(ROOT, "isaac_amd/assets/xbot_model.json"), "w") as f:
        json.dump(model, f, indent=1)
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.5f" % b["mass"], b.get("joint"), b.get("axis"), "rot" in b)
    print("total mass %.5f" % model["total_mass"])


def main():
    bodies = collapse(URDF)
    assert len(bodies) == 11, len(bodies)
    total = sum(b["mass"] for b in bodies)

    # collision primitives: list of (body index, [points in body frame])
    name_to_idx = {b["name"]: i for i, b in enumerate(bodies)}
    contacts = leg_and_base_contacts(name_to_idx)

    model = {"source": "resources/robots/hector_v2/xacro/robot.urdf (collapse_fixed_joints)",
             "total_mass": total, "bodies": bodies, "contacts": contacts}
    os.makedirs(os.path.join(ROOT, "isaac_amd/assets"), exist_ok=True)
    with open(os.path.join(ROOT, "isaac_amd/assets/hector_model.json"), "w") as f:
        json.dump(model, f, indent=1)

    # ---- C header ----
    L = []
    L.append("// GENERATED by tools/compile_urdf.py from the reference's robot.urdf -- do not edit.")
    L.append("// 11 collapsed bodies / 10 revolute joints; body i>0 is driven by joint i-1.")
    L.append("#pragma once")
    L.append("#define HX_NB 11")
    L.append("#define HX_NJ 10")

    def flit(v):
        t = "%.9g" % v
        if "." not in t and "e" not in t and "n" not in t:
            t += ".0"
        return t + "f"

    def arr(name, vals):
        L.append("static constexpr float %s[%d] = {%s};" % (name, len(vals), ", ".join(flit(v) for v in vals)))

    def iarr(name, vals):
        L.append("static constexpr int %s[%d] = {%s};" % (name, len(vals), ", ".join(str(v) for v in vals)))

    iarr("HXM_PARENT", [b["parent"] for b in bodies])
    iarr("HXM_AXIS", [b.get("axis", -1) for b in bodies])
    arr("HXM_MASS", [b["mass"] for b in bodies])
    arr("HXM_COM", [x for b in bodies for x in b["com"]])
    arr("HXM_ICOM", [np.array(b["inertia_com"])[i, j] for b in bodies
                     for (i, j) in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))])
    # spatial inertia about the body-frame origin: I = [[Io, hx],[hx^T, m 1]], h = m*com,
    # Io = Ic + m (c.c 1 - c c^T)   (6 values: xx yy zz xy xz yz)
    io = []
    for b in bodies:
        c = np.array(b["com"]); m = b["mass"]
        Io = np.array(b["inertia_com"]) + m * (c @ c * np.eye(3) - np.outer(c, c))
        io += [Io[0, 0], Io[1, 1], Io[2, 2], Io[0, 1], Io[0, 2], Io[1, 2]]
    arr("HXM_IO", io)
    arr("HXM_H", [b["mass"] * x for b in bodies for x in b["com"]])
    arr("HXM_OFFSET", [x for b in bodies for x in b.get("offset", [0, 0, 0])])
    arr("HXM_QLO", [b["lower"] for b in bodies[1:]])
    arr("HXM_QHI", [b["upper"] for b in bodies[1:]])
    arr("HXM_VMAX", [b["velocity"] for b in bodies[1:]])
    arr("HXM_EFFORT", [b["effort"] for b in bodies[1:]])
    iarr("HXM_CONTACT_BODY", [c["body"] for c in contacts])
    arr("HXM_CONTACT_PTS", [x for c in contacts for p in c["points"] for x in p])
    L.append("#define HX_NCSHAPE %d" % len(contacts))
    # per-leg constant table staged in LDS by the two-lanes-per-robot kernel:
    # leg l, local body b (0 hip, 1 hip2, 2 thigh, 3 calf, 4 toe) at [l*128 + b*16 + ...]:
    #   0-2 joint offset, 3-5 h = m*com, 6-11 Io (xx yy zz xy xz yz), 12 mass, 13 q_lo, 14 q_hi, 15 v_max
    # then 24 floats thigh box corners at [l*128 + 80], 24 floats toe box corners at [l*128 + 104]
    legc = []
    for leg in range(2):
        row = []
        for b in range(5):
            bd = bodies[1 + leg * 5 + b]
            c = np.array(bd["com"]); m = bd["mass"]
            Io = np.array(bd["inertia_com"]) + m * (c @ c * np.eye(3) - np.outer(c, c))
            row += list(bd["offset"]) + list(m * c) + [Io[0, 0], Io[1, 1], Io[2, 2], Io[0, 1], Io[0, 2], Io[1, 2]]
            row += [m, bd["lower"], bd["upper"], bd["velocity"]]
        row += [x for pt in contacts[1 + 2 * leg]["points"] for x in pt]
        row += [x for pt in contacts[2 + 2 * leg]["points"] for x in pt]
        assert len(row) == 128
        legc += row
    L.append("#define HX_LEGC_STRIDE 128")
    L.append("__device__ static const float HXM_LEGC[256] = {%s};" % ", ".join(flit(v) for v in legc))
    with open(os.path.join(ROOT, "isaac_amd/csrc/hx_model_data.h"), "w") as f:
        f.write("\n".join(L) + "\n")

    print("bodies:")
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.5f" % b["mass"], "com", np.round(b["com"], 4),
              b.get("joint"), b.get("offset"), b.get("axis"))
    print("total mass %.5f" % total)
    for c in contacts:
        print("contact", bodies[c["body"]]["name"], c.get("bbox"))


if __name__ == "__main__":
    sys.exit(main_full() if "--full" in sys.argv else main_xbot() if "--xbot" in sys.argv else main())
