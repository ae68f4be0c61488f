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

Collision geometry (PhysX collides the convex hull of every collision mesh; a body's shape here is a set of points on
that hull, tested against the ground):
  * toes: foot_L2.stl / foot_R2.stl are present -- 12 support points of their convex hull (the sole's two ends, the
    rounded tips and the top edge, on both side faces of the blade-shaped foot);
  * base: body.stl is absent -> the trunk box of const.xacro:17-19; plus the six arm meshes that exist
    (ShoulderYaw / ShoulderPitch / ForeArmPitch, both sides; robot.urdf fixes the arms to the trunk, so with
    collapse_fixed_joints their shapes belong to the base -- they count for terminate_after_contacts_on=['base', ...]),
    8 hull support points each; UpperArmRoll_*.stl is absent (no shape), the head's collision is a 1 mm box (ignored);
  * hips: L_hip1.STL / L_hip2.STL are absent; hip.STL and hip2_L/R.STL of the same directory (simplified solids of these
    two links, bounding boxes 8 x 5 x 6.5 cm and 7.5 x 5.5 x 5 cm) stand in, 8 hull support points each;
  * thighs: thigh_combined_*2.stl is absent -> the box of const.xacro:128-133;  calves: no collision in the URDF.
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


def stl_points(path):
    """All triangle vertices of a binary or ASCII STL, [n, 3]."""
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
    return np.asarray(pts, np.float64)


def stl_bbox(path):
    """Axis-aligned bounding box of an STL."""
    pts = stl_points(path)
    return pts.min(0), pts.max(0)


DIAG8 = [(sx, sy, sz) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]


def hull_support(pts, dirs):
    """Support points of the convex hull of `pts` in the given directions (vertices of the hull; duplicates removed,
    order of first appearance) -- an inner approximation of the hull that is exact along those directions."""
    out, seen = [], set()
    for d in dirs:
        i = int(np.argmax(pts @ np.asarray(d, np.float64)))
        key = tuple(np.round(pts[i], 6))
        if key not in seen:
            seen.add(key)
            out.append(pts[i].tolist())
    return out


def foot_points(path):
    """12 points of the foot hull.  The foot is a blade: a 2-D profile in x-z (flat sole 14 cm long at z = -0.04, tips
    rounded with ~9 mm radius, ridge at z = +0.0125 above the ankle) extruded 1.8 cm in y.  Per side face: the two ends of
    the flat sole, the two tip points, the two ends of the ridge."""
    pts = stl_points(path)
    out = []
    for ysel in (-1.0, 1.0):
        e = 1e-3 * ysel
        out += hull_support(pts, [(-0.08, e, -1.0), (0.08, e, -1.0), (-1.0, e, -0.35), (1.0, e, -0.35), (-0.2, e, 1.0), (0.2, e, 1.0)])
    assert len(out) == 12, len(out)
    return out


def collapse(urdf_path, sort_children=False, link_poses=None):
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
        if link_poses is not None:
            link_poses[name] = (len(bodies), p.copy(), R.copy())      # the body being built gets index len(bodies)
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


def leg_and_base_contacts(name_to_idx, fixed_arms=False):
    """Collision point sets of the biped proper (same for robot.urdf and robot_w_arm.urdf): see the module docstring.
    fixed_arms: robot.urdf, whose arm links are fixed to the trunk -- their meshes become shapes of the base."""
    contacts = [{"body": name_to_idx["base"], "points": box_corners((0, 0, 0), (0.125, 0.19, 0.248)),
                 "source": "const.xacro:17-19 trunk box (body.stl absent)"}]
    if fixed_arms:
        contacts += fixed_arm_contacts(name_to_idx["base"])
    for side, sgn in (("L", 1.0), ("R", -1.0)):
        for link, mesh in ((f"{side}_hip", "hip.STL"), (f"{side}_hip2", f"hip2_{side}.STL")):
            pts = stl_points(os.path.join(MESH_DIR, mesh))
            if link.endswith("_hip") and side == "R":
                pts = pts * np.array([1.0, -1.0, 1.0])        # one hip.STL for both sides: mirror it for the right leg
            contacts.append({"body": name_to_idx[link], "points": hull_support(pts, DIAG8),
                             "source": f"{mesh} convex-hull support points (stand-in for the absent {side}_hip{'1' if link.endswith('_hip') else '2'}.STL)"})
        contacts.append({"body": name_to_idx[f"{side}_thigh"],
                         "points": box_corners((0, sgn * 0.0175, -0.09), (0.06, 0.035, 0.18)),
                         "source": "const.xacro:128-133 thigh box (thigh_combined_*2.stl absent)"})
        path = os.path.join(MESH_DIR, f"foot_{side}2.stl")
        lo, hi = stl_bbox(path)
        contacts.append({"body": name_to_idx[f"{side}_toe"], "points": foot_points(path),
                         "source": f"foot_{side}2.stl convex-hull support points", "bbox": [lo.tolist(), hi.tolist()]})
    return contacts


def fixed_arm_contacts(base_idx):
    """robot.urdf:594-890: eight arm links hang off the trunk through FIXED joints.  Their collision meshes, placed by the
    chain of joint origins, as 8 hull support points each in the base frame."""
    root = ET.parse(URDF).getroot()
    joints = {j.find("child").get("link"): j for j in root.findall("joint")}
    out = []
    for link in root.findall("link"):
        nm = link.get("name")
        if nm.split("_")[-1] not in ("twist", "shoulder", "roll", "elbow"):
            continue
        col = link.find("collision")
        mesh = col.find("geometry").find("mesh") if col is not None and col.find("geometry") is not None else None
        if mesh is None:
            continue
        path = os.path.normpath(os.path.join(os.path.dirname(URDF), mesh.get("filename")))
        if not os.path.exists(path):
            continue
        chain, l = [], nm
        while l in joints:
            assert joints[l].get("type") == "fixed" or joints[l].find("parent").get("link") == "base", nm
            chain.append(joints[l]); l = joints[l].find("parent").get("link")
        p, R = np.zeros(3), np.eye(3)
        for j in reversed(chain):
            jp, jR = parse_origin(j.find("origin"))
            p, R = p + R @ jp, R @ jR
        co, cR = parse_origin(col.find("origin"))
        pts = (stl_points(path) @ (R @ cR).T) + (p + R @ co)
        out.append({"body": base_idx, "points": hull_support(pts, DIAG8),
                    "source": f"{nm}: {os.path.basename(path)} convex-hull support points, fixed to the trunk"})
    assert len(out) == 6, [o["source"] for o in out]
    return out


def emit_model_header(prefix, struct, title, bodies, contacts, nl, chains, path):
    """C tables of one robot for the kernels and the host build (isaac_amd/csrc/hx_dyn.h reads them through a model
    descriptor).  Per body side (left, then right) one table of `SIDE_STRIDE` floats:
        joints  [nl][JSTRIDE]: 0-2 joint offset in the parent frame, 3-5 h = m * com, 6-11 inertia about the body-frame
                origin (xx yy zz xy xz yz), 12 mass, 13 q_lo, 14 q_hi, 15 v_max, and for robots with rotated joint frames
                16-24 the constant child -> parent rotation (row-major)
        shapes  per body with collision points, in body order: bounding sphere (centre xyz, radius), then the points
    and one base table: 2 x BASE_NSUB sub-shapes (the two lanes of a robot take BASE_NSUB each), [sphere 4][BASE_NP points]
    each, then inertia (6), h (3), mass."""
    by_body, base_subs = {}, []
    for c in contacts:
        if c["body"] == 0:
            base_subs.append(c["points"])          # the base keeps its shapes apart: each gets its own bounding sphere
        else:
            by_body.setdefault(c["body"], []).extend(c["points"])
    has_rot = any("rot" in b for b in bodies)
    jstride = 28 if has_rot else 16

    def Io(bd):
        c = np.array(bd["com"]); m = bd["mass"]
        I = np.array(bd["inertia_com"]) + m * (c @ c * np.eye(3) - np.outer(c, c))
        return [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    def sphere(pts):
        pts = np.array(pts)
        ctr = (pts.min(0) + pts.max(0)) / 2
        return list(ctr) + [float(np.sqrt(((pts - ctr) ** 2).sum(1)).max())]

    def flit(v):
        t = "%.9g" % v
        if "." not in t and "e" not in t and "n" not in t:
            t += ".0"
        return t + "f"

    npts = [len(by_body.get(1 + k, [])) for k in range(nl)]
    assert npts == [len(by_body.get(1 + nl + k, [])) for k in range(nl)], "both sides must carry the same point counts"
    pts_off, off = [], nl * jstride
    for k in range(nl):
        pts_off.append(off if npts[k] else -1)
        off += (4 + 3 * npts[k]) if npts[k] else 0
    side_stride = (off + 3) // 4 * 4
    side = []
    for sd in range(2):
        row = []
        for k in range(nl):
            bd = bodies[1 + sd * nl + k]
            j = list(bd["offset"]) + list(bd["mass"] * np.array(bd["com"])) + Io(bd) + [bd["mass"], bd["lower"], bd["upper"], bd["velocity"]]
            if has_rot:
                j += list(np.array(bd.get("rot", np.eye(3))).reshape(-1)) + [0.0, 0.0, 0.0]
            assert len(j) == jstride
            row += j
        for k in range(nl):
            pts = by_body.get(1 + sd * nl + k, [])
            if pts:
                row += sphere(pts) + [x for pt in pts for x in pt]
        row += [0.0] * (side_stride - len(row))
        side += row
    # base sub-shapes: BASE_NP points each, BASE_NSUB per lane (lane 0 takes the first half of the list, lane 1 the rest;
    # an odd count is padded with an empty sub-shape: radius -1e9 never passes the bounding-sphere test)
    base_np = max(len(x) for x in base_subs)
    assert all(len(x) == base_np for x in base_subs), "base sub-shapes must carry the same number of points"
    nsub = (len(base_subs) + 1) // 2
    blocks = [sphere(x) + [v for pt in x for v in pt] for x in base_subs]
    while len(blocks) < 2 * nsub:
        blocks.insert(nsub if len(base_subs) > 1 else len(blocks), [0.0, 0.0, 0.0, -1.0e9] + [0.0] * (3 * base_np))
    for k in range(nl):
        if npts[k]:
            assert sphere(by_body[1 + k])[3] <= 0.4, "side shapes must fit the 0.4 m reach of the 3 x 3 terrain pool entries (hx_dyn.h shape_gap)"
    base = [v for blk in blocks for v in blk] + Io(bodies[0]) + list(bodies[0]["mass"] * np.array(bodies[0]["com"])) + [bodies[0]["mass"]]
    axis = [bodies[1 + k]["axis"] for k in range(nl)]
    assert axis == [bodies[1 + nl + k]["axis"] for k in range(nl)]
    ints = lambda v: ", ".join(str(x) for x in v)
    L = [f"// GENERATED by tools/compile_urdf.py from the reference's {title} -- do not edit.",
         "#pragma once",
         f"struct {struct} {{",
         f"  static constexpr int NL = {nl};                 // joints per body side",
         f"  static constexpr int NCH = {len(chains)};                // kinematic chains per side, each hanging off the base",
         f"  static constexpr int CH_START[{len(chains)}] = {{{ints(c[0] for c in chains)}}};",
         f"  static constexpr int CH_LEN[{len(chains)}] = {{{ints(c[1] for c in chains)}}};",
         f"  static constexpr int AXIS[{nl}] = {{{ints(axis)}}};",
         f"  static constexpr bool HAS_ROT = {'true' if has_rot else 'false'};",
         f"  static constexpr int JSTRIDE = {jstride};",
         f"  static constexpr int NPTS[{nl}] = {{{ints(npts)}}};        // collision points per side-local body",
         f"  static constexpr int PTS_OFF[{nl}] = {{{ints(pts_off)}}};   // float offset of the body's [sphere 4][points 3 * n] block",
         f"  static constexpr int SIDE_STRIDE = {side_stride};",
         f"  static constexpr int BASE_NSUB = {nsub};             // base sub-shapes per lane, [sphere 4][points 3 * BASE_NP] each",
         f"  static constexpr int BASE_NP = {base_np};",
         f"  static constexpr int BASE_FLOATS = {len(base)};",
         "};",
         f"HX_TABLE float {prefix}_SIDE[{len(side)}] = {{{', '.join(flit(v) for v in side)}}};",
         f"HX_TABLE float {prefix}_BASE[{len(base)}] = {{{', '.join(flit(v) for v in base)}}};",
         f"static constexpr float {prefix}_MASS0 = {flit(bodies[0]['mass'])};",
         f"static constexpr float {prefix}_EFFORT[{2 * nl}] = {{{', '.join(flit(b['effort']) for b in bodies[1:])}}};"]
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")


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
        contacts.append({"body": name_to_idx[nm], "points": hull_support(stl_points(path) + o, DIAG8),
                         "source": os.path.basename(path) + " convex-hull support points", "bbox": [lo.tolist(), hi.tolist()]})
    model = {"source": "resources/robots/hector_v2/xacro/robot_w_arm.urdf (collapse_fixed_joints, children in alphabetical order)",
             "total_mass": sum(b["mass"] for b in bodies), "bodies": bodies, "contacts": contacts}
    with open(os.path.join(ROOT, "isaac_amd/assets/hector_full_model.json"), "w") as f:
        json.dump(model, f, indent=1)
    emit_model_header("HXF", "HXM_Full", "robot_w_arm.urdf (19 bodies / 18 revolute joints in Isaac Gym order: L leg, L arm, R leg, R arm)",
                      bodies, contacts, 9, [(0, 5), (5, 4)], os.path.join(ROOT, "isaac_amd/csrc/hx_model_data_full.h"))
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.5f" % b["mass"], b.get("joint"), b.get("axis"))
    print("total mass %.5f" % model["total_mass"])


def link_collision_points(urdf, link_poses, name):
    """Vertices of link `name`'s collision mesh in the frame of the collapsed body that carries it (None: no mesh / file)."""
    root = ET.parse(urdf).getroot()
    link = root.find(f"link[@name='{name}']")
    col = link.find("collision") if link is not None else None
    mesh = col.find("geometry").find("mesh") if col is not None and col.find("geometry") is not None else None
    if mesh is None:
        return None
    path = os.path.normpath(os.path.join(os.path.dirname(urdf), mesh.get("filename")))
    if not os.path.exists(path):          # two hand meshes are referenced as ./meshes/... in the URDF
        path = os.path.normpath(os.path.join(os.path.dirname(urdf), "..", "meshes", os.path.basename(mesh.get("filename"))))
    if not os.path.exists(path):
        return None
    _, p, R = link_poses[name]
    co, cR = parse_origin(col.find("origin"))
    return (stl_points(path) @ (R @ cR).T) + (p + R @ co)


def main_xbot():
    """XBot-L (task humanoid_ppo): model data for the oracle only.  Twelve revolute joints about the local z
    of frames rotated against their parents ("rot"), upper body and hands collapsed into the base."""
    urdf = os.path.join(REF, "resources/robots/XBot/urdf/XBot-L.urdf")
    poses = {}
    bodies = collapse(urdf, link_poses=poses)
    assert len(bodies) == 13, len(bodies)
    assert [b.get("joint") for b in bodies[1:]] == [f"{s}_{j}_joint" for s in ("left", "right") for j in
                                                    ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll")]
    name_to_idx = {b["name"]: i for i, b in enumerate(bodies)}
    # Collision geometry = what XBot-L.urdf enables (round 2): the base_link box; everything fixed to the base -- head,
    # both arms with their hands -- as three groups of 8 hull support points; per leg the thigh (leg_pitch_link), the
    # calf (knee_link) and the foot (ankle_roll_link) meshes, 8 hull support points each, in the frames of their collapsed
    # bodies (a body turned by a negative joint axis carries its mesh turned with it: link_poses holds that turn).
    contacts = [{"body": 0, "points": box_corners((0, 0, 0.1), (0.4, 0.4, 0.4)), "source": "XBot-L.urdf base_link collision box"}]
    groups = {"head": [], "left arm": [], "right arm": []}
    for nm, (bi, _, _) in poses.items():
        if bi != 0 or nm == "base_link":
            continue
        pts = link_collision_points(urdf, poses, nm)
        if pts is None:
            continue
        groups["left arm" if nm.startswith("left_") else "right arm" if nm.startswith("right_") else "head"].append(pts)
    for g in ("head", "left arm", "right arm"):
        assert groups[g], g
        contacts.append({"body": 0, "points": hull_support(np.concatenate(groups[g]), DIAG8),
                         "source": f"{g}: {len(groups[g])} collision meshes fixed to base_link, convex-hull support points of their union"})
    for side in ("left", "right"):
        for link in ("leg_pitch_link", "knee_link", "ankle_roll_link"):
            nm = f"{side}_{link}"
            pts = link_collision_points(urdf, poses, nm)
            assert pts is not None and poses[nm][0] == name_to_idx[nm], nm
            contacts.append({"body": name_to_idx[nm], "points": hull_support(pts, DIAG8),
                             "source": nm + ".STL convex-hull support points"})
    model = {"source": "resources/robots/XBot/urdf/XBot-L.urdf (collapse_fixed_joints)",
             "total_mass": sum(b["mass"] for b in bodies), "bodies": bodies, "contacts": contacts}
    with open(os.path.join(ROOT, "isaac_amd/assets/xbot_model.json"), "w") as f:
        json.dump(model, f, indent=1)
    emit_model_header("HXX", "HXM_XBot", "XBot-L.urdf (13 bodies / 12 revolute joints about the z axes of rotated joint frames)",
                      bodies, contacts, 6, [(0, 6)], os.path.join(ROOT, "isaac_amd/csrc/hx_model_data_xbot.h"))
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.5f" % b["mass"], b.get("joint"), b.get("axis"), "rot" in b)
    print("total mass %.5f" % model["total_mass"])


def main():
    bodies = collapse(URDF)
    assert len(bodies) == 11, len(bodies)
    total = sum(b["mass"] for b in bodies)

    # collision primitives: list of (body index, [points in body frame])
    name_to_idx = {b["name"]: i for i, b in enumerate(bodies)}
    contacts = leg_and_base_contacts(name_to_idx, fixed_arms=True)

    model = {"source": "resources/robots/hector_v2/xacro/robot.urdf (collapse_fixed_joints)",
             "total_mass": total, "bodies": bodies, "contacts": contacts}
    os.makedirs(os.path.join(ROOT, "isaac_amd/assets"), exist_ok=True)
    with open(os.path.join(ROOT, "isaac_amd/assets/hector_model.json"), "w") as f:
        json.dump(model, f, indent=1)

    emit_model_header("HXM", "HXM_Hector", "robot.urdf (11 collapsed bodies / 10 revolute joints; body i > 0 is driven by joint i - 1)",
                      bodies, contacts, 5, [(0, 5)], os.path.join(ROOT, "isaac_amd/csrc/hx_model_data.h"))

    print("bodies:")
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.5f" % b["mass"], "com", np.round(b["com"], 4),
              b.get("joint"), b.get("offset"), b.get("axis"))
    print("total mass %.5f" % total)
    for c in contacts:
        print("contact", bodies[c["body"]]["name"], c.get("bbox"))


if __name__ == "__main__":
    sys.exit(main_full() if "--full" in sys.argv else main_xbot() if "--xbot" in sys.argv else main())
