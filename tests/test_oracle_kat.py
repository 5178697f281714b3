"""CPU tests (no GPU): known-answer vectors that pin the oracle.

The reference's own tests assert nothing (src/test.rs), so these vectors are hand-derived from the reference
source; each case says which lines it pins.  PARITY UNPINNED by the reference itself (oracle/oracle.h)."""
import ctypes as C

import numpy as np

from oracle import oracle_py as O

SPHERE, TRI = 0, 1
LAMB = [(0, 0, (0.5, 0.5, 0.5), 0.0)]


def scene(prims):
    p = np.zeros(len(prims), O.PRIM_DT)
    cen = np.zeros((len(prims), 3))
    for i, (t, v) in enumerate(prims):
        p[i]["type"] = t; p[i]["material"] = 0; p[i]["v"][:len(v)] = v
        cen[i] = v[:3] if t == SPHERE else (np.array(v[0:3]) + np.array(v[3:6]) + np.array(v[6:9])) / 3.0
    return O.OracleScene(p, cen, np.array(LAMB, O.MAT_DT))


def test_philox_random123_vectors():
    """Philox4x32-10 against the Random123 kat_vectors (the RNG replaces thread_rng(), rtweekend.rs:166-169)."""
    def ph(c, k):
        out = (C.c_uint32 * 4)()
        O.lib().orc_philox4x32_10((C.c_uint32 * 4)(*c), (C.c_uint32 * 2)(*k), out)
        return list(out)
    assert ph([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rng_draw_is_53_bits_of_the_block():
    out = (C.c_uint32 * 4)()
    O.lib().orc_philox4x32_10((C.c_uint32 * 4)(7, 3, 2, 0), (C.c_uint32 * 2)(5, 0), out)
    lo = ((out[1] << 32) | out[0]) >> 11
    hi = ((out[3] << 32) | out[2]) >> 11
    assert O.rng_draw(5, 7, 3, 2, 0) == lo * 2.0 ** -53      # rand 0.8 Standard f64
    assert O.rng_draw(5, 7, 3, 2, 1) == hi * 2.0 ** -53
    assert 0.0 <= O.rng_draw(5, 7, 3, 2, 5) < 1.0


def test_sphere_roots_exact():
    """sphere.rs:28-47: a=1, h=1, c=0.75, disc=0.25 -> roots 0.5 and 1.5 exactly; interval is strict."""
    sc = scene([(SPHERE, [0, 0, -1, 0.5])])
    r = sc.trace("naive", [[0, 0, 0]], [[0, 0, -1]])
    assert r["hit"][0] and r["prim"][0] == 0 and r["t"][0] == 0.5 and r["tp"][0] == 0.5
    r = sc.trace("naive", [[0, 0, 0]], [[0, 0, -1]], tmin=0.5)          # surrounds() is strict: 0.5 rejected, far root taken
    assert r["hit"][0] and r["t"][0] == 1.5
    r = sc.trace("naive", [[0, 0, 0]], [[0, 0, -1]], tmin=0.5, tmax=1.5)
    assert not r["hit"][0] and r["prim"][0] == -1
    r = sc.trace("naive", [[0, 0, 0]], [[0, 0, -2]])                    # direction is not normalised: t halves
    assert r["t"][0] == 0.25


def test_triangle_is_single_sided_and_interval_tests_n_dot_d():
    """triangle.rs:39-47.  triangle_test.json tri 0 has n = (-0.3, 0.4, -3.0); a ray from the origin to its centroid has
    n.d = 6.9 > 0 and hits at t = 1; the same triangle wound the other way has n.d < 0 and is invisible;
    and the interval's upper end is compared with n.d, not with t."""
    v0, v1, v2 = [-1.0, 0.0, -2.2], [-3.0, 0.0, -2.0], [-2.0, 1.5, -1.9]
    d = [-2.0, 0.5, -(2.2 + 2.0 + 1.9) / 3.0]
    sc = scene([(TRI, v0 + v1 + v2)])
    r = sc.trace("naive", [[0, 0, 0]], [d])
    assert r["hit"][0] and r["prim"][0] == 0 and abs(r["t"][0] - 1.0) < 1e-14
    assert not scene([(TRI, v0 + v2 + v1)]).trace("naive", [[0, 0, 0]], [d])["hit"][0]
    assert not sc.trace("naive", [[0, 0, 0]], [d], tmax=5.0)["hit"][0]      # t = 1 < 5, but n.d = 6.9 is not < 5
    assert sc.trace("naive", [[0, 0, 0]], [d], tmax=7.0)["hit"][0]
    assert not sc.trace("naive", [[0, 0, 0]], [d], tmin=6.95)["hit"][0]     # and the lower end too


A = (TRI, [-1, -1, -1, -1, 1, -1, 3, -1, -1])            # n = (0,0,-8); hit at t = 1 for o=0, d=(0,0,-1)
B = (TRI, [9, -0.1, -3, 9, 0.1, -3, 9.4, -0.1, -3])      # n = (0,0,-0.08); plane at t = 3, ray misses the triangle


def test_failed_triangle_test_clobbers_rec_t_and_p():
    """triangle.rs:52,59: rec.t and rec.p are written before the edge tests.  A then B (file order): B's n.d = 0.08
    lies inside (0.001, closest=1), its plane distance 3 overwrites rec.t and rec.p, its edge tests fail; the record
    keeps A's normal/material.  B then A: A is accepted last and nothing clobbers it."""
    r = scene([A, B]).trace("naive", [[0, 0, 0]], [[0, 0, -1]])
    assert r["hit"][0] and r["prim"][0] == 0
    assert abs(r["t"][0] - 3.0) < 1e-12 and abs(r["tp"][0] - 3.0) < 1e-12
    r = scene([B, A]).trace("naive", [[0, 0, 0]], [[0, 0, -1]])
    assert r["hit"][0] and r["prim"][0] == 1 and r["t"][0] == 1.0 and r["tp"][0] == 1.0


def test_last_accepted_triangle_wins_not_the_closest():
    """A farther triangle tested later is accepted if its n.d (not its t) is below closest_so_far (triangle.rs:44-57)."""
    far = (TRI, [-0.1, -0.1, -5, -0.1, 0.1, -5, 0.3, -0.1, -5])     # n = (0,0,-0.08): n.d = 0.08 < 1; t = 5; ray inside it
    r = scene([A, far]).trace("naive", [[0, 0, 0]], [[0, 0, -1]])
    assert r["hit"][0] and r["prim"][0] == 1 and abs(r["t"][0] - 5.0) < 1e-12


def test_bvh_equals_naive_on_spheres():
    """Spheres accept only roots inside (tmin, closest) and write the record only on success (sphere.rs:41-53), so every
    traversal order returns the closest hit: naive and the three BVH flavours must agree bit for bit."""
    rng = np.random.default_rng(0)
    prims = [(SPHERE, list(rng.uniform(-8, 8, 3)) + [float(rng.uniform(0.2, 1.5))]) for _ in range(60)]
    o = rng.uniform(-12, 12, (4000, 3)); d = rng.normal(size=(4000, 3))
    sc = scene(prims)
    ref = sc.trace("naive", o, d)
    assert ref["hit"].sum() > 500
    for mode in ("sah-position", "sah-plane", "naive"):
        got = sc.build_bvh(mode).trace("bvh", o, d)
        assert np.array_equal(got["hit"], ref["hit"]) and np.array_equal(got["prim"], ref["prim"])
        assert np.array_equal(got["t"][ref["hit"]].view(np.uint64), ref["t"][ref["hit"]].view(np.uint64))


def test_bvh_counters_small_case():
    """One sphere: root box test (bvh.rs:183) + one leaf visit (bvh.rs:190) + one primitive test (bvh.rs:197)."""
    sc = scene([(SPHERE, [0, 0, -1, 0.5])]).build_bvh()
    st = sc.trace("bvh", [[0, 0, 0]], [[0, 0, -1]])["stats"]
    assert (st["box_tests"], st["prim_tests"], st["traversal_steps"], st["intersection_checks"]) == (1, 1, 1, 2)
    st = sc.trace("bvh", [[0, 0, 0]], [[0, 0, 1]])["stats"]              # misses the root box: hit() is never entered
    assert (st["box_tests"], st["prim_tests"], st["traversal_steps"]) == (1, 0, 0)


def test_bvh_node_numbering_and_leaf_order():
    """bvh.rs:167-175: left = ptr+1, right = ptr+2, left subtree numbered first; bvh.rs:136-140: the slice is sorted by
    the best axis.  Four unit spheres on the x axis: SAH splits {0,1}|{2,3}, then each pair."""
    sc = scene([(SPHERE, [x, 0, 0, 0.4]) for x in (3.0, 0.0, 2.0, 1.0)]).build_bvh("sah-position")
    nodes, order = sc.get_bvh()
    assert len(nodes) == 7
    assert not nodes[0]["is_leaf"] and (nodes[0]["left"], nodes[0]["right"]) == (1, 2)
    assert (nodes[1]["left"], nodes[1]["right"]) == (3, 4) and (nodes[2]["left"], nodes[2]["right"]) == (5, 6)
    assert list(order) == [1, 3, 2, 0]
    assert [int(nodes[i]["first"]) for i in (3, 4, 5, 6)] == [0, 1, 2, 3]
    assert np.allclose(nodes[0]["min"], [-0.4, -0.4, -0.4]) and np.allclose(nodes[0]["max"], [3.4, 0.4, 0.4])


def test_grid_quirks():
    """grid.rs:24-27,42-45: the table is filled with clones of the origin cell and cells within distance 1 of the origin are
    skipped, so at size 1 the three axis neighbours of cell 0 keep the origin cell's box."""
    sc = scene([(SPHERE, [0, 0, -1, 0.5])]).build_grid(1.0)
    aabb, off, pl = sc.get_grid()
    assert len(aabb) == 8_000_000
    for idx in (0, 1, 200, 40000):
        assert list(aabb[idx]) == [-100, -100, -100, -99, -99, -99]
    assert list(aabb[2]) == [-98, -100, -100, -97, -99, -99]
    # sphere box [-0.5,0.5]x[-0.5,0.5]x[-1.5,-0.5] overlaps (inclusive, aabb.rs:113-117) the unit cells with
    # min x,y in {-1,0} and min z in {-2,-1}: 8 cells, one entry each
    assert off[-1] == len(pl) and len(pl) == 2 * 2 * 2


def test_camera_initialise_known_values():
    """camera.rs:88-128 for the default preset (main.rs:98-103): vfov 90, focus 1 -> viewport 2 high; 900x506."""
    cam = O.camera(16.0 / 9.0, 900, 90.0, (0, 0, 0), (0, 0, -1), (0, 1, 0), 0.1, 1.0)
    assert (cam.image_width, cam.image_height) == (900, 506)
    vh = 2.0 * np.tan(np.radians(90.0) / 2.0)
    assert abs(cam.dv[1] + vh / 506) < 1e-15 and abs(cam.du[0] - vh * (900 / 506) / 900) < 1e-15
    assert list(cam.center) == [0, 0, 0] and abs(cam.pixel00[2] + 1.0) < 1e-15


def test_resolve_gamma_clamp_quantise():
    """color.rs:6-30: sqrt gamma, clamp to [0, 0.999], 256x truncation; camera.rs:75 scale by 1/spp."""
    s = np.array([[0.0, -1.0, 4.0], [1.0, 0.25 * 4, 0.01 * 4]]) 
    rgb = O.resolve(s, 4)
    assert rgb.tolist() == [[0, 0, 255], [128, 128, 25]]
