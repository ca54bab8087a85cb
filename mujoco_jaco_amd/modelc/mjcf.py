"""MJCF -> flat "raw model" arrays (the subset of MJCF the Jaco models use).

Input data: /root/reference/env_script/assets/jaco2/*.xml + meshes/*.STL (read-only
model *data* of the reference).  The array naming follows MuJoCo's public mjModel
field names so the oracle reads like the algorithm descriptions it restates [EXT].
MuJoCo defaults that the reference relies on because its XML has no <option>
 are spelled out in DEFAULTS below.
"""
import os
import re
from xml.etree import ElementTree

import numpy as np

from . import mesh as meshmod
from . import rot

GEOM_PLANE, GEOM_SPHERE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 0, 2, 5, 6, 7
GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH}
JNT_FREE, JNT_HINGE = 0, 3

DEFAULTS = dict(
    timestep=0.002, gravity=(0.0, 0.0, -9.81), tolerance=1e-8, iterations=100,
    mpr_tolerance=1e-6, mpr_iterations=50,
    geom_friction=(1.0, 0.005, 0.0001), solref=(0.02, 1.0), solimp=(0.9, 0.95, 0.001, 0.5, 2.0),
    density=1000.0, condim=3, contype=1, conaffinity=1, margin=0.0, gap=0.0,
)


_NUM = re.compile(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?")


def _numbers(s):
    """Numeric attribute -> list of floats the way MuJoCo reads it: stream extraction (`istringstream >> double`), i.e. each number is the
    longest valid prefix and the next one starts right behind it -- white space is not required between them.  The reference's own
    files rely on that: jaco2_torque.xml:47 has size=".01 .02.035", which MuJoCo reads as (.01, .02, .035)."""
    out, pos = [], 0
    s = s.strip()
    while pos < len(s):
        while pos < len(s) and s[pos].isspace():
            pos += 1
        if pos >= len(s):
            break
        m = _NUM.match(s, pos)
        if m is None:
            raise ValueError("not a number at %r in attribute %r" % (s[pos:pos + 8], s))
        out.append(float(m.group(0)))
        pos = m.end()
    return out


def _floats(s, n=None, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array(_numbers(s), dtype=np.float64)
    if n is not None and len(v) < n and default is not None:
        full = np.array(default, dtype=np.float64)
        full[: len(v)] = v
        return full
    return v


def _frame(el):
    pos = _floats(el.get("pos"), default=(0, 0, 0))
    if el.get("quat") is not None:
        quat = rot.quat_normalize(_floats(el.get("quat")))
    elif el.get("euler") is not None:
        quat = rot.euler_xyz_quat(_floats(el.get("euler")))
    elif el.get("zaxis") is not None:
        # [EXT] MuJoCo: the minimal rotation that takes (0, 0, 1) to the given axis (jaco2_curtain_torque_sensor.xml:73: the curtain rod)
        z = _floats(el.get("zaxis")); z = z / np.linalg.norm(z)
        ax = np.cross([0.0, 0.0, 1.0], z); sn, cs = np.linalg.norm(ax), z[2]
        if sn < 1e-12:
            quat = np.array([1.0, 0, 0, 0]) if cs > 0 else np.array([0.0, 1.0, 0, 0])
        else:
            half = 0.5 * np.arctan2(sn, cs)
            quat = np.concatenate([[np.cos(half)], np.sin(half) * ax / sn])
    else:
        quat = np.array([1.0, 0, 0, 0])
    return pos, quat


class _Body:
    pass


def _expand_includes(el, xml_dir, depth=0):
    """<include file="..."/>: MuJoCo splices the children of the included file's root element in at that position (the dual-arm
    model pulls its two arms in this way, jaco2_dual_torque.xml:48-49)."""
    if depth > 8:
        raise ValueError("include nesting too deep")
    i = 0
    while i < len(el):
        ch = el[i]
        if ch.tag == "include":
            inc = ElementTree.parse(os.path.join(xml_dir, ch.get("file"))).getroot()
            _expand_includes(inc, xml_dir, depth + 1)
            el.remove(ch)
            for k, sub in enumerate(list(inc)):
                el.insert(i + k, sub)
            i += len(inc)
        else:
            _expand_includes(ch, xml_dir, depth)
            i += 1


def parse(xml_path, timestep=None):
    """Parse one MJCF file into a dict of numpy arrays + name lists."""
    root = ElementTree.parse(xml_path).getroot()
    xml_dir = os.path.dirname(xml_path)
    _expand_includes(root, xml_dir)
    comp = root.find("compiler")
    meshdir = comp.get("meshdir", "") if comp is not None else ""
    assert comp is None or comp.get("angle", "degree") == "radian"

    # ---- mesh assets actually instantiated by a geom (dice.STL is declared but unused and
    # missing from the checkout: jaco2_curtain_torque.xml:30, .MISSING_LARGE_BLOBS:1)
    used = {g.get("mesh") for g in root.iter("geom") if g.get("mesh")}
    mesh_names, meshes = [], []
    for m in root.findall("asset/mesh"):
        fn = m.get("file")
        name = m.get("name") or os.path.splitext(os.path.basename(fn))[0]
        if name not in used:
            continue
        scale = _floats(m.get("scale"), default=(1, 1, 1))
        mesh_names.append(name)
        meshes.append(meshmod.process_mesh(os.path.join(xml_dir, meshdir, fn), scale))

    bodies, joints, geoms, sites = [], [], [], []
    body_names, joint_names, geom_names, site_names = [], [], [], []

    def add_geom(el, bid):
        g = {}
        g["type"] = GEOM_TYPES[el.get("type", "sphere")]
        g["body"] = bid
        pos, quat = _frame(el)
        g["size"] = _floats(el.get("size"), 3, (0, 0, 0))
        g["dataid"] = -1
        density = float(el.get("density", DEFAULTS["density"]))
        if g["type"] == GEOM_MESH:
            mi = mesh_names.index(el.get("mesh"))
            mm = meshes[mi]
            g["dataid"] = mi
            pos = pos + rot.quat_to_mat(quat) @ mm["pos"]
            quat = rot.quat_normalize(rot.quat_mul(quat, mm["quat"]))
            g["mass"] = density * mm["volume"]
            g["inertia"] = density * mm["inertia"]
            g["rbound"] = mm["rbound"]
            g["size"] = mm["aabb"].copy()  # half-extents of the hull in the geom frame (culling only)
        elif g["type"] == GEOM_BOX:
            sx, sy, sz = g["size"]
            g["mass"] = density * 8 * sx * sy * sz
            g["inertia"] = g["mass"] / 3 * np.array([sy * sy + sz * sz, sx * sx + sz * sz, sx * sx + sy * sy])
            g["rbound"] = float(np.linalg.norm(g["size"]))
        elif g["type"] == GEOM_SPHERE:
            r = g["size"][0]
            g["mass"] = density * 4 / 3 * np.pi * r ** 3
            g["inertia"] = np.full(3, 0.4 * g["mass"] * r * r)
            g["rbound"] = float(r)
        elif g["type"] == GEOM_CYLINDER:   # size = (radius, half height), axis = local z
            r, h = g["size"][0], g["size"][1]
            g["mass"] = density * np.pi * r * r * 2 * h
            g["inertia"] = g["mass"] * np.array([r * r / 4 + h * h / 3, r * r / 4 + h * h / 3, r * r / 2])
            g["rbound"] = float(np.hypot(r, h))
        elif g["type"] == GEOM_PLANE:
            g["mass"], g["inertia"], g["rbound"] = 0.0, np.zeros(3), 0.0
        else:
            raise NotImplementedError(el.get("type"))
        g["pos"], g["quat"] = pos, quat
        g["contype"] = int(el.get("contype", DEFAULTS["contype"]))
        g["conaffinity"] = int(el.get("conaffinity", DEFAULTS["conaffinity"]))
        g["condim"] = int(el.get("condim", DEFAULTS["condim"]))
        g["friction"] = _floats(el.get("friction"), 3, DEFAULTS["geom_friction"])
        g["solref"] = _floats(el.get("solref"), 2, DEFAULTS["solref"])
        g["solimp"] = _floats(el.get("solimp"), 5, DEFAULTS["solimp"])
        g["margin"] = float(el.get("margin", DEFAULTS["margin"]))
        g["gap"] = float(el.get("gap", DEFAULTS["gap"]))
        geoms.append(g)
        geom_names.append(el.get("name", ""))

    def add_body(el, parent):
        b = _Body()
        bid = len(bodies)
        bodies.append(b)
        body_names.append(el.get("name", "world") if parent >= 0 else "world")
        b.parent = max(parent, 0)
        if parent < 0:
            b.pos, b.quat = np.zeros(3), np.array([1.0, 0, 0, 0])
        else:
            b.pos, b.quat = _frame(el)
        b.mocap = el.get("mocap", "false") == "true"
        b.joints, b.geoms = [], []
        inert = el.find("inertial")
        b.explicit = inert is not None
        if b.explicit:
            b.ipos = _floats(inert.get("pos"), default=(0, 0, 0))
            b.mass = float(inert.get("mass"))
            d = _floats(inert.get("diaginertia"))
            iq = rot.quat_normalize(_floats(inert.get("quat"))) if inert.get("quat") else np.array([1.0, 0, 0, 0])
            R = rot.quat_to_mat(iq)
            b.inertia = R @ np.diag(d) @ R.T
        for ch in el:
            if ch.tag in ("joint", "freejoint"):
                j = {"body": bid, "name": ch.get("name", "")}
                if ch.tag == "freejoint" or ch.get("type") == "free":
                    j["type"] = JNT_FREE
                    j["pos"], j["axis"] = np.zeros(3), np.array([0.0, 0, 1])
                else:
                    assert ch.get("type", "hinge") == "hinge"
                    j["type"] = JNT_HINGE
                    j["pos"] = _floats(ch.get("pos"), default=(0, 0, 0))
                    ax = _floats(ch.get("axis"), default=(0, 0, 1))
                    j["axis"] = ax / np.linalg.norm(ax)
                j["ref"] = float(ch.get("ref", 0.0))
                j["limited"] = ch.get("limited", "false") == "true"
                j["range"] = _floats(ch.get("range"), default=(0, 0))
                j["damping"] = float(ch.get("damping", 0.0))
                j["stiffness"] = float(ch.get("stiffness", 0.0))      # joint spring (jaco2_torque.xml:109-133: the distal finger joints)
                j["springref"] = float(ch.get("springref", 0.0))
                j["solref"] = _floats(ch.get("solreflimit"), 2, DEFAULTS["solref"])
                j["solimp"] = _floats(ch.get("solimplimit"), 5, DEFAULTS["solimp"])
                b.joints.append(len(joints))
                joints.append(j)
                joint_names.append(j["name"])
            elif ch.tag == "geom":
                b.geoms.append(ch)
            elif ch.tag == "site":
                s = {"body": bid}
                s["pos"], s["quat"] = _frame(ch)
                s["type"] = GEOM_TYPES[ch.get("type", "sphere")]
                s["size"] = _floats(ch.get("size"), 3, (0.005, 0.005, 0.005))
                sites.append(s)
                site_names.append(ch.get("name", ""))
        for ch in el:
            if ch.tag == "body":
                add_body(ch, bid)

    add_body(root.find("worldbody"), -1)
    # geoms are stored grouped by body, in body order
    for bid, b in enumerate(bodies):
        for gel in b.geoms:
            add_geom(gel, bid)

    nbody = len(bodies)
    # ---- inertial properties of bodies without <inertial>: from their geoms
    for bid, b in enumerate(bodies):
        if b.explicit:
            continue
        gs = [g for g in geoms if g["body"] == bid]
        mass = sum(g["mass"] for g in gs)
        if bid == 0 or mass <= 0:
            b.mass, b.ipos, b.inertia = 0.0, np.zeros(3), np.zeros((3, 3))
            continue
        com = sum(g["mass"] * g["pos"] for g in gs) / mass
        I = np.zeros((3, 3))
        for g in gs:
            R = rot.quat_to_mat(g["quat"])
            d = g["pos"] - com
            I += R @ np.diag(g["inertia"]) @ R.T + g["mass"] * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        b.mass, b.ipos, b.inertia = mass, com, I

    # ---- indices
    nq = nv = 0
    body_jntadr, body_jntnum, body_dofadr, body_dofnum = [], [], [], []
    for b in bodies:
        body_jntadr.append(b.joints[0] if b.joints else -1)
        body_jntnum.append(len(b.joints))
        body_dofadr.append(nv if b.joints else -1)
        nd = 0
        for ji in b.joints:
            j = joints[ji]
            j["qposadr"], j["dofadr"] = nq, nv
            if j["type"] == JNT_FREE:
                nq, nv, nd = nq + 7, nv + 6, nd + 6
            else:
                nq, nv, nd = nq + 1, nv + 1, nd + 1
        body_dofnum.append(nd)
    weld = []
    for bid, b in enumerate(bodies):
        weld.append(bid if (b.joints or bid == 0) else weld[b.parent])
    mocapid, nmocap = [], 0
    for b in bodies:
        if b.mocap:
            mocapid.append(nmocap)
            nmocap += 1
        else:
            mocapid.append(-1)

    qpos0 = np.zeros(nq)
    dof_bodyid, dof_jntid, dof_damping = [], [], []
    for ji, j in enumerate(joints):
        if j["type"] == JNT_FREE:
            b = bodies[j["body"]]
            qpos0[j["qposadr"]:j["qposadr"] + 3] = b.pos
            qpos0[j["qposadr"] + 3:j["qposadr"] + 7] = b.quat
            n = 6
        else:
            qpos0[j["qposadr"]] = j["ref"]
            n = 1
        dof_bodyid += [j["body"]] * n
        dof_jntid += [ji] * n
        dof_damping += [j["damping"]] * n
    dof_parentid = []
    for d in range(nv):
        bid = dof_bodyid[d]
        if d > body_dofadr[bid]:
            dof_parentid.append(d - 1)
            continue
        p = bodies[bid].parent
        while p > 0 and body_dofnum[p] == 0:
            p = bodies[p].parent
        dof_parentid.append(body_dofadr[p] + body_dofnum[p] - 1 if p > 0 else -1)

    # ---- actuators / sensors
    act = []
    for a in root.find("actuator") if root.find("actuator") is not None else []:
        ji = joint_names.index(a.get("joint"))
        d = {"jnt": ji, "position": a.tag == "position", "kp": float(a.get("kp", 1.0))}
        d["ctrllimited"] = a.get("ctrllimited", "false") == "true"
        d["ctrlrange"] = _floats(a.get("ctrlrange"), default=(0, 0))
        d["forcelimited"] = a.get("forcelimited", "false") == "true"
        d["forcerange"] = _floats(a.get("forcerange"), default=(0, 0))
        assert a.tag in ("motor", "position"), a.tag
        act.append(d)
    sensor_site, sensor_names = [], []
    for s in root.find("sensor") if root.find("sensor") is not None else []:
        assert s.tag == "touch"
        sensor_site.append(site_names.index(s.get("site")))
        sensor_names.append(s.get("name"))

    hull = [m["hull_vert"] for m in meshes]
    vertadr = np.cumsum([0] + [len(h) for h in hull])[:-1] if hull else np.zeros(0)

    f64 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.float64).reshape(-1))
    i32 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.int32).reshape(-1))
    M = {
        "nq": i32([nq]), "nv": i32([nv]), "nu": i32([len(act)]), "nbody": i32([nbody]), "njnt": i32([len(joints)]),
        "ngeom": i32([len(geoms)]), "nsite": i32([len(sites)]), "nmesh": i32([len(meshes)]), "nmocap": i32([nmocap]),
        "nsensor": i32([len(sensor_site)]),
        "opt_timestep": f64([DEFAULTS["timestep"] if timestep is None else timestep]),
        "opt_gravity": f64(DEFAULTS["gravity"]), "opt_tolerance": f64([DEFAULTS["tolerance"]]),
        "opt_iterations": i32([DEFAULTS["iterations"]]),
        "opt_mpr_tolerance": f64([DEFAULTS["mpr_tolerance"]]), "opt_mpr_iterations": i32([DEFAULTS["mpr_iterations"]]),
        "body_parentid": i32([b.parent for b in bodies]), "body_weldid": i32(weld), "body_mocapid": i32(mocapid),
        "body_jntadr": i32(body_jntadr), "body_jntnum": i32(body_jntnum),
        "body_dofadr": i32(body_dofadr), "body_dofnum": i32(body_dofnum),
        "body_pos": f64([b.pos for b in bodies]), "body_quat": f64([b.quat for b in bodies]),
        "body_ipos": f64([b.ipos for b in bodies]), "body_inertia": f64([b.inertia for b in bodies]),
        "body_mass": f64([b.mass for b in bodies]),
        "jnt_type": i32([j["type"] for j in joints]), "jnt_bodyid": i32([j["body"] for j in joints]),
        "jnt_qposadr": i32([j["qposadr"] for j in joints]), "jnt_dofadr": i32([j["dofadr"] for j in joints]),
        "jnt_pos": f64([j["pos"] for j in joints]), "jnt_axis": f64([j["axis"] for j in joints]),
        "jnt_limited": i32([j["limited"] for j in joints]), "jnt_range": f64([j["range"] for j in joints]),
        "jnt_solref": f64([j["solref"] for j in joints]), "jnt_solimp": f64([j["solimp"] for j in joints]),
        "qpos0": f64(qpos0),
        "dof_bodyid": i32(dof_bodyid), "dof_jntid": i32(dof_jntid), "dof_parentid": i32(dof_parentid),
        "dof_damping": f64(dof_damping),
        # joint springs: qfrc_passive -= stiffness * (qpos - springref) on hinge joints (free joints carry none in these models)
        "jnt_stiffness": f64([j["stiffness"] if j["type"] == JNT_HINGE else 0.0 for j in joints]),
        "jnt_springref": f64([j["springref"] if j["type"] == JNT_HINGE else 0.0 for j in joints]),
        "geom_type": i32([g["type"] for g in geoms]), "geom_bodyid": i32([g["body"] for g in geoms]),
        "geom_dataid": i32([g["dataid"] for g in geoms]), "geom_contype": i32([g["contype"] for g in geoms]),
        "geom_conaffinity": i32([g["conaffinity"] for g in geoms]), "geom_condim": i32([g["condim"] for g in geoms]),
        "geom_pos": f64([g["pos"] for g in geoms]), "geom_quat": f64([g["quat"] for g in geoms]),
        "geom_size": f64([g["size"] for g in geoms]), "geom_rbound": f64([g["rbound"] for g in geoms]),
        "geom_friction": f64([g["friction"] for g in geoms]), "geom_solref": f64([g["solref"] for g in geoms]),
        "geom_solimp": f64([g["solimp"] for g in geoms]), "geom_margin": f64([g["margin"] for g in geoms]),
        "geom_gap": f64([g["gap"] for g in geoms]),
        "mesh_vertadr": i32(vertadr), "mesh_vertnum": i32([len(h) for h in hull]),
        "mesh_vert": f64(np.concatenate(hull) if hull else np.zeros((0, 3))),
        "site_bodyid": i32([s["body"] for s in sites]), "site_type": i32([s["type"] for s in sites]),
        "site_pos": f64([s["pos"] for s in sites]), "site_quat": f64([s["quat"] for s in sites]),
        "site_size": f64([s["size"] for s in sites]),
        "actuator_jntid": i32([a["jnt"] for a in act]), "actuator_position": i32([a["position"] for a in act]),
        "actuator_kp": f64([a["kp"] for a in act]), "actuator_ctrllimited": i32([a["ctrllimited"] for a in act]),
        "actuator_ctrlrange": f64([a["ctrlrange"] for a in act]),
        "actuator_forcelimited": i32([a["forcelimited"] for a in act]),
        "actuator_forcerange": f64([a["forcerange"] for a in act]),
        "sensor_siteid": i32(sensor_site),
        "mocap_pos0": f64([b.pos for b in bodies if b.mocap]), "mocap_quat0": f64([b.quat for b in bodies if b.mocap]),
    }
    names = {"body": body_names, "joint": joint_names, "geom": geom_names, "site": site_names,
             "sensor": sensor_names, "mesh": mesh_names}
    return M, names
