import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gym_novel_gridworlds_amd import VecNovelGridworld, make_spec, LidarInFront
from oracle.ngw_oracle import Oracle, lidar
spec = make_spec('NovelGridworld-Pogostick-v1', 10)
A = len(spec.actions_id)
for n2, H, dt in ((8192, 12, np.int32), (8192, 12, np.int16), (8192, 100, np.int32), (8200, 12, np.int32)):
    v = VecNovelGridworld(spec=spec, num_envs=n2, device=0, seed=7, autoreset=True, horizon=H)
    w = LidarInFront(v, num_beams=8, dtype=dt)
    o = Oracle(spec.compile(), n2, seed=7, autoreset=True, horizon=H)
    cc = w._lidar.compile(spec)
    w.reset(); o.reset()
    rs = np.random.RandomState(0)
    for t in range(2 * H + 6 if H < 50 else 110):
        a = rs.randint(0, A, size=n2).astype(np.int32)
        obs, reward, done, info = w.step(a); o.step(a)
        exp = lidar(cc, 10, len(spec.items_id), o.st.map, o.st.loc, o.st.facing, o.st.inv)
        bad = np.nonzero((obs != exp).any(1))[0]
        if bad.size:
            dev = v.lidar_observation(device=True).cpu().numpy()
            bd = np.nonzero((dev != exp).any(1))[0]
            print('n %d H %d %s step %d: host rows wrong for %d envs (first %s), device rows wrong for %d; done of first bad %s; wrong cols %s' % (n2, H, np.dtype(dt).name, t, bad.size, bad[:5], bd.size, done[bad[0]], np.nonzero(obs[bad[0]] != exp[bad[0]])[0][:10]))
            print('   host', obs[bad[0]][:16], '\n   exp ', exp[bad[0]][:16], '\n   dev ', dev[bad[0]][:16])
            break
    else:
        print('n %d H %d %s: ok' % (n2, H, np.dtype(dt).name))
    v.close()
