"""Oscillator-network controller: oracle known answers on the CPU, HIP kernel vs oracle on the GPU."""
import numpy as np
import pytest


def _net(freq, rate, amp, con, out):
    from farms_mujoco_amd.control import OscillatorNetwork
    return OscillatorNetwork(freq, rate, amp, con, out)


def test_single_oscillator_known_answer(oracle):
    """No coupling: the phase advances by 2 pi f h per step exactly; the amplitude follows the critically damped
    second-order law r'' = a (a/4 (R - r) - r') towards R; output = r (1 + cos theta)."""
    net = _net([2.0], [20.0], [0.5], [], [(0, -1, 1.0, 0.25)])
    h, T = 1e-3, 2000
    tape, ph, amp, damp = oracle.cpg_tape(net, T, h, np.zeros((1, 1)), np.zeros((1, 1)), np.zeros((1, 1)))
    th = 2*np.pi*2.0*h*T
    th = (th + np.pi) % (2*np.pi) - np.pi
    assert abs(((ph[0, 0] - th + np.pi) % (2*np.pi)) - np.pi) < 1e-9
    t = h*T                                             # closed form r(t) = R (1 - (1 + a t / 2) exp(-a t / 2))
    r_exact = 0.5*(1 - (1 + 10*t)*np.exp(-10*t))
    assert abs(amp[0, 0] - r_exact) < 2e-3              # explicit Euler at h = 1e-3
    assert np.allclose(tape[0, 0, 0], 0.25)             # r = 0 at the first step
    k = 1500
    # the tape holds the state before each step: recompute from a shorter run
    _, phk, ampk, _ = oracle.cpg_tape(net, k, h, np.zeros((1, 1)), np.zeros((1, 1)), np.zeros((1, 1)))
    assert np.isclose(tape[k, 0, 0], 0.25 + ampk[0, 0]*(1 + np.cos(phk[0, 0])), atol=1e-12)


def test_two_oscillators_lock_to_the_phase_bias(oracle):
    phi = 0.7
    net = _net([1.0, 1.0], [20.0, 20.0], [1.0, 1.0], [(1, 0, 5.0, phi), (0, 1, 5.0, -phi)], [(0, 1, 1.0, 0.0)])
    ph0 = np.array([[0.3, -1.1]])
    _, ph, amp, _ = oracle.cpg_tape(net, 6000, 1e-3, ph0, np.ones((1, 2)), np.zeros((1, 2)))
    d = (ph[0, 0] - ph[0, 1] - phi + np.pi) % (2*np.pi) - np.pi        # sin(th_0 - th_1 - phi) -> 0
    assert abs(d) < 1e-6 and np.allclose(amp, 1.0, atol=1e-9)


def test_salamander_network_travelling_wave(oracle):
    """The double chain settles into a head-to-tail travelling wave with the requested total lag."""
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.control import salamander_network
    m = salamander33()
    net = salamander_network(m, n_wave=1.0)
    assert net.n_osc == 2*11 + 2*4 and net.nu == m.nu
    rng = np.random.default_rng(0)
    ph0 = net.initial_phase[None, :] + rng.uniform(-0.3, 0.3, (1, net.n_osc))
    tape, ph, amp, _ = oracle.cpg_tape(net, 8000, 1e-3, ph0, np.zeros((1, net.n_osc)), np.zeros((1, net.n_osc)))
    left = ph[0, 0:22:2]
    lag = (left[:-1] - left[1:] + np.pi) % (2*np.pi) - np.pi
    assert np.allclose(lag, 2*np.pi/11, atol=0.05), lag
    pos = [a for a in range(m.nu) if m.actuator_tags[a] == 'position' and 'body' in m.actuator_names[a]]
    swing = tape[-1000:, 0, pos].max(0) - tape[-1000:, 0, pos].min(0)
    assert np.all(swing > 0.5) and np.all(swing < 0.7)                 # 2 * 2 * 0.15 = 0.6 rad peak to peak


@pytest.mark.gpu
def test_hip_tape_matches_oracle(oracle):
    import torch
    from farms_mujoco_amd.model import salamander33, synthetic_batch
    from farms_mujoco_amd.control import salamander_network, NetworkController
    m = salamander33()
    n, T = 16, 500
    _, _, psi = synthetic_batch(m, n)
    net = salamander_network(m)
    drive = 1.0 + 0.1*np.arange(n)/n
    c = NetworkController(m, net, n, env_phase=psi, drive=drive)
    ph0, a0, d0 = (x.cpu().numpy().astype(np.float64) for x in (c.phase, c.amp, c.damp))
    tape = c.ctrl_tape(T).cpu().numpy()
    ref, ph, amp, damp = oracle.cpg_tape(net, T, m.timestep, ph0, a0, d0, drive=drive.astype(np.float32))
    assert np.abs(tape - ref).max() < 2e-4, np.abs(tape - ref).max()
    assert np.abs(c.amp.cpu().numpy() - amp).max() < 1e-5
    dphi = (c.phase.cpu().numpy() - ph + np.pi) % (2*np.pi) - np.pi
    assert np.abs(dphi).max() < 2e-4
    # chunking is invisible: two calls of T/2 give the same tape
    c2 = NetworkController(m, net, n, env_phase=psi, drive=drive)
    t2 = torch.cat([c2.ctrl_tape(T//2).clone(), c2.ctrl_tape(T - T//2).clone()]).cpu().numpy()
    assert np.array_equal(t2, tape)
