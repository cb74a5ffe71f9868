"""The MJCF exporter cannot rot: every case of the MuJoCo pin (tests/mujoco_pin.py) exports to XML that parses and holds what
the model holds - element counts, the body nesting, the contact pairs, the heightfield and mesh assets, the keyframe.  No MuJoCo
needed; the comparison of the numbers themselves is tests/test_vs_mujoco.py (skips without ``mujoco``).
Reference: what Simulation.save_mjcf_xml writes (simulation.py:215-225), from the model mjcf.py:1174-1512 builds."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import mujoco_pin as mp


@pytest.mark.parametrize('name', list(mp.CASES))
@pytest.mark.parametrize('fusestatic', [True, False])
def test_export_holds_the_model(name, fusestatic):
    from farms_mujoco_amd.simulation.mjcf import model2mjcf_xml
    from farms_mujoco_amd.model import GEOM_MESH, GEOM_HFIELD
    m = mp.case_model(name)
    root = ET.fromstring(model2mjcf_xml(m, fusestatic=fusestatic))
    assert root.find('compiler').get('fusestatic') == ('true' if fusestatic else 'false')
    bodies = root.find('worldbody').iter('body')
    names = [b.get('name') for b in bodies]
    assert names == list(m.body_names[1:])                      # document order = depth-first pre-order = body ids
    parent = {c.get('name'): p.get('name') for p in root.find('worldbody').iter('body') for c in p.findall('body')}
    for b in range(1, m.nbody):
        want = m.body_names[int(m.body_parentid[b])] if m.body_parentid[b] > 0 else None
        assert parent.get(m.body_names[b]) == want
    joints = list(root.find('worldbody').iter('joint')) + list(root.find('worldbody').iter('freejoint'))
    assert len(joints) == m.njnt and sum(1 for j in joints if j.get('limited') == 'true') == int(np.sum(np.asarray(m.jnt_limited) != 0))
    geoms = list(root.find('worldbody').iter('geom'))
    assert len(geoms) == m.ngeom and sorted(g.get('name') for g in geoms) == sorted(f'geom_{g}' for g in range(m.ngeom))
    # animat geoms never collide with each other through the masks: only through explicit pairs
    for g in geoms:
        gi = int(g.get('name')[5:])
        assert g.get('conaffinity') == ('1' if m.geom_bodyid[gi] == 0 else '0') and g.get('contype') == '1' and g.get('condim') == '3'
    pairs = root.find('contact').findall('pair') if root.find('contact') is not None else []
    assert len(pairs) == int(getattr(m, 'npair', 0))
    for p_, e in enumerate(pairs):
        assert e.get('geom1') == f'geom_{int(m.pair_geom1[p_])}' and e.get('geom2') == f'geom_{int(m.pair_geom2[p_])}' and e.get('condim') == '3'
    asset = root.find('asset')
    n_mesh = int(np.sum(np.asarray(m.geom_type) == GEOM_MESH)); n_hf = int(np.sum(np.asarray(m.geom_type) == GEOM_HFIELD))
    assert (len(asset.findall('mesh')) if asset is not None else 0) == n_mesh
    assert (len(asset.findall('hfield')) if asset is not None else 0) == n_hf
    if n_hf:
        hf = asset.find('hfield')
        elev = np.array(hf.get('elevation').split(), float)
        assert elev.size == int(m.hfield_nrow)*int(m.hfield_ncol) == int(hf.get('nrow'))*int(hf.get('ncol'))
        assert elev.min() >= 0.0 and elev.max() <= 1.0
        # the surface the XML describes is the model's: elevation * size z + geom z offset == data * hfield_size[2] + geom pos z
        size = np.array(hf.get('size').split(), float)
        ge = [g for g in geoms if g.get('type') == 'hfield'][0]
        gi = int(ge.get('name')[5:])
        zoff = float(ge.get('pos').split()[2]) - float(m.geom_pos[gi][2])
        assert np.allclose(elev*size[2] + zoff, np.asarray(m.hfield_data, float).ravel()*float(m.hfield_size[2]), atol=1e-12)
    for g in geoms:
        if g.get('type') == 'mesh':
            gi = int(g.get('name')[5:])
            vert = np.array(asset.find(f"mesh[@name='{g.get('mesh')}']").get('vertex').split(), float).reshape(-1, 3)
            a0, n = int(m.geom_vertadr[gi]), int(m.geom_vertnum[gi])
            assert np.array_equal(vert, np.asarray(m.mesh_vert[a0:a0 + n], float))
    act = root.find('actuator')
    assert (len(list(act)) if act is not None else 0) == m.nu
    sens = root.find('sensor')
    assert [s.get('name') for s in sens] == m.sensor_names()
    # sensordata width the sensors imply = the model's layout (framelinvel / frameangvel: 3 each; the rest scalars)
    assert sum(3 if s.tag in ('framelinvel', 'frameangvel') else 1 for s in sens) == m.nsensordata
    key = root.find('keyframe').find('key')
    assert len(key.get('qpos').split()) == m.nq and len(key.get('qvel').split()) == m.nv
    opt = root.find('option')
    assert float(opt.get('timestep')) == m.timestep and int(opt.get('iterations')) == m.solver_iterations
    assert opt.get('solver') == {0: 'PGS', 1: 'CG', 2: 'Newton'}[int(m.solver)] and opt.get('cone') == {0: 'pyramidal', 1: 'elliptic'}[int(m.cone)]


def test_pin_inputs_are_deterministic_and_fp32_representable(oracle):
    """The inputs the MuJoCo fixtures are generated from are reproducible from the case name alone and exactly representable in
    fp32, so that the GPU path, the oracle and MuJoCo start from bit-identical numbers."""
    for name in ('salamander33_swim', 'salamander33_walk_pgs', 'tree_contacts_100'):
        m = mp.case_model(name)
        a = mp.case_inputs(name, m, oracle); b = mp.case_inputs(name, mp.case_model(name), oracle)
        for k in a:
            assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], a[k].astype(np.float32).astype(np.float64)), (name, k)
        assert a['qpos'].shape == (mp.N_ENVS, m.nq)
    m = mp.case_model('salamander33_walk_pgs')
    w = mp.case_inputs('salamander33_walk_pgs', m, oracle)
    o = oracle.step_tf(m, w['qpos'], w['qvel'], ctrl=w['ctrl'], want_AR=False)
    assert (o['ncon'] > 0).all(), 'the walking pin states must stand on the ground'
