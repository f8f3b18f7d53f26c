import sys, time; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle, HUMANOID_HBM
np.set_printoptions(precision=5, suppress=True, linewidth=200)
m = hb.Model.load(HUMANOID_HBM)
print('sizes', m.nq, m.nv, m.nu, m.nobs)
N = 8
b = hb.Batch(m, N, 0)
b.diag_enable(True)
o = Oracle()
# teacher forced: same state on both, one step, compare
b.reset(perturb=True)
st = b.get_state(hb.STATE_INTEGRATION, dtype=np.float64)
ctrl = np.zeros((N, m.nu), np.float32)
for e in range(N): ctrl[e] = o.ctrl_env(0, e)
b.step(ctrl)
q1 = b.qpos; v1 = b.qvel
nc, ne, ni = b.counts()
print('gpu ncon', nc, 'nefc', ne, 'niter', ni, 'status', b.status())
for e in range(N):
    o.reset()
    o.qpos[:] = st[e, 1:1+m.nq]; o.qvel[:] = st[e,1+m.nq:1+m.nq+m.nv]; o.qacc_warmstart[:] = st[e,1+m.nq+m.nv:]
    o.ctrl[:] = ctrl[e]
    o.step()
    print(e, 'oracle ncon', o.ncon, 'nefc', o.nefc, 'dq', np.abs(q1[e]-o.qpos).max(), 'dv', np.abs(v1[e]-o.qvel).max(), 'vmax', np.abs(o.qvel).max())
# free-running 200 steps vs oracle env 0
b.reset(perturb=True)
o.init_env(0)
for t in range(300):
    for e in range(N): ctrl[e] = o.ctrl_env(t, e)
    b.step(ctrl)
    o.ctrl[:] = ctrl[0]; o.step()
    if t % 30 == 0:
        q = b.qpos[0]
        print(t, 'drift', np.abs(q - o.qpos).max(), 'z', q[2], o.qpos[2], 'nefc', b.counts()[1][0], o.nefc)
print('status', b.status())
# timing
N = 4096
b2 = hb.Batch(m, N, 0)
b2.reset(perturb=True)
b2.rollout_halton(10); b2.sync()
t0 = time.time(); b2.rollout_halton(100, t0=10); b2.sync(); dt = time.time()-t0
print('rollout 100 steps x 4096: %.4f s -> %.3e env-steps/s' % (dt, N*100/dt))
import ctypes
t0 = time.time()
for i in range(100): b2.rollout_halton(1, t0=110+i)
b2.sync(); dt = time.time()-t0
print('100 single-step launches: %.4f s -> %.3e env-steps/s' % (dt, N*100/dt))
nc, ne, ni = b2.counts(); print('mean ncon', nc.mean(), 'nefc', ne.mean(), 'niter', ni.mean(), 'max nefc', ne.max(), 'status nonzero', (b2.status()!=0).sum())
b3 = hb.Batch(m, N, 0)
b3.reset(perturb=True); b3.pipeline(True)  # two segments
b3.rollout_halton(10); b3.sync()
b3.rollout_halton(100, t0=10); b3.sync()
t0 = time.time()
for i in range(100): b3.rollout_halton(1, t0=110+i)
b3.sync(); dt = time.time()-t0
print('100 single-step launches, pipelined: %.4f s -> %.3e env-steps/s' % (dt, N*100/dt))
s2 = b2.get_state(hb.STATE_INTEGRATION); s3 = b3.get_state(hb.STATE_INTEGRATION)
print('pipelined state identical:', np.array_equal(s2, s3))
