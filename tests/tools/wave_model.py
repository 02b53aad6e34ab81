"""CPU model of the compositing kernel at wave granularity (C2 recipe, 48 random tiles of one view): trips of phase 1 (longest
of the four 16-lane group lists) and phase 2 (largest per-lane pass count per 64-entry window) per tile-list entry, and
the lane utilisation of both. Needs the oracle (test infrastructure); not used by the product path."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import make_scene, run_oracle


def bbox_px(v2g, thr, W, H, fx, fy):
    v = v2g.astype(np.float64); C = v[:,9]
    k = -2.0*thr.astype(np.float64) + 4e-6*np.abs(C) + 1e-3
    cK = C - k
    B0,B1,B2 = v[:,6],v[:,7],v[:,8]
    m00 = cK*v[:,0]-B0*B0; m01 = cK*v[:,1]-B0*B1; m02 = cK*v[:,2]-B0*B2
    m11 = cK*v[:,3]-B1*B1; m12 = cK*v[:,4]-B1*B2; m22 = cK*v[:,5]-B2*B2
    D00 = m11*m22-m12*m12; D11 = m00*m22-m02*m02; D22 = m00*m11-m01*m01
    D02 = m01*m12-m02*m11; D12 = m01*m02-m00*m12
    with np.errstate(all="ignore"):
        cx = D02/D22; cy = D12/D22
        hx = np.sqrt(D02*D02-D00*D22)/np.abs(D22); hy = np.sqrt(D12*D12-D11*D22)/np.abs(D22)
    ok = (cK > 0) & (m00 > 0) & (D22 > 1e-12*np.abs(m00*m11)) & np.isfinite(cx) & np.isfinite(cy) & np.isfinite(hx) & np.isfinite(hy)
    hx = hx*1.02 + 0.75/fx; hy = hy*1.02 + 0.75/fy
    x0 = (cx-hx)*fx + W/2 - 0.5; x1 = (cx+hx)*fx + W/2 - 0.5; y0 = (cy-hy)*fy + H/2-0.5; y1 = (cy+hy)*fy+H/2-0.5
    big = 1e30
    x0 = np.where(ok, x0, -big); x1 = np.where(ok, x1, big); y0 = np.where(ok, y0, -big); y1 = np.where(ok, y1, big)
    return np.stack([x0,x1,y0,y1],1), ok


sc = make_scene(P=196608, res=(256,256), s0=0.01, view="oblique"); o = run_oracle(sc)
W=H=256; f32=np.float32
fx = float(f32(W)/(f32(2.0)*f32(sc["tanfovx"])))
v2g = o["view2gaussian"]; opac = o["conic_opacity"][:,3]
thr = (np.log(f32(1.0)/(f32(255.0)*opac)).astype(f32) - f32(1e-4))
bb, ok = bbox_px(v2g, thr, W, H, fx, fx)
ranges, pl = o["ranges"], o["point_list"]; nc = o["n_contrib"][0]
rng = np.random.default_rng(0); tiles = rng.choice(256, 48, replace=False)
v64 = v2g.astype(np.float64)
R = 255
stats = dict(entries=0, rounds=0, p1_trips=0, p1_lane=0, p2_trips=0, p2_lane=0, waves_rounds=0, q_trips=0, p2_trips_round=0)
for tile in tiles:
    r0,r1 = ranges[tile]; ids = pl[r0:r1]; n=len(ids)
    ty, tx = divmod(tile, 16)
    ys, xs = np.meshgrid(np.arange(ty*16, ty*16+16), np.arange(tx*16, tx*16+16), indexing="ij")
    rx = ((xs+0.5-128)/fx).reshape(-1,1); ry = ((ys+0.5-128)/fx).reshape(-1,1)
    v = v64[ids][None]
    n0 = v[...,0]*rx + v[...,1]*ry + v[...,2]; n1 = v[...,1]*rx+v[...,3]*ry+v[...,4]; n2 = v[...,2]*rx+v[...,4]*ry+v[...,5]
    a = rx*n0+ry*n1+n2; b = v[...,6]*rx+v[...,7]*ry+v[...,8]
    p = -0.5*(v[...,9] - b*b/a)
    hit = (opac[ids][None]*np.exp(np.minimum(p,0)) >= 1/255).reshape(16,16,n)
    last = nc[ty*16:ty*16+16, tx*16:tx*16+16].astype(np.int64)   # last contributor (1-based count)
    # done index: first hit strictly after last contributor (the entry that trips T<1e-4), else n
    doneidx = np.full((16,16), n, dtype=np.int64)
    for y in range(16):
        for x in range(16):
            h = np.nonzero(hit[y,x,last[y,x]:])[0]
            if len(h): doneidx[y,x] = last[y,x] + h[0]      # processing this entry sets done
    b_ = bb[ids]
    stats["entries"] += n
    for r in range(0, n, R):
        e0, e1 = r, min(n, r+R)
        if (doneidx < e0).all(): break
        stats["rounds"] += 1
        for wv in range(4):
            qx, qy = wv & 1, wv >> 1
            counts = []; lists = []
            for g in range(4):
                sx, sy = g & 1, g >> 1
                px0, py0 = tx*16 + qx*8 + sx*4, ty*16 + qy*8 + sy*4
                sel = (b_[e0:e1,0] <= px0+3) & (b_[e0:e1,1] >= px0) & (b_[e0:e1,2] <= py0+3) & (b_[e0:e1,3] >= py0)
                lists.append(np.nonzero(sel)[0] + e0)
            cmax = max(len(l) for l in lists)
            stats["waves_rounds"] += 1
            # union (quadrant list) length for reference
            stats["q_trips"] += len(np.unique(np.concatenate(lists)))
            p2round = np.zeros(64, dtype=np.int64)
            for w0 in range(0, cmax, 64):
                wn = min(64, cmax - w0)
                # lanes
                lane_pass = np.zeros(64, dtype=np.int64); lane_tests = np.zeros(64, dtype=np.int64); alive = np.zeros(64, bool)
                li = 0
                for g in range(4):
                    sx, sy = g & 1, g >> 1
                    l = lists[g][w0:w0+64]
                    for iy in range(4):
                        for ix in range(4):
                            y, x = qy*8+sy*4+iy, qx*8+sx*4+ix
                            lane = g*16 + iy*4 + ix
                            d = doneidx[y,x]
                            if len(l) == 0 or d < (l[0] if len(l) else 0):
                                # done before window start (approx): lane inactive in phase 1 if done
                                pass
                            act = d >= (lists[g][w0] if w0 < len(lists[g]) else 10**9)
                            alive[lane] = act
                            if act:
                                lane_tests[lane] = len(l)
                                hh = hit[y,x,l] & (l <= d)
                                lane_pass[lane] = hh.sum()
                if alive.any():
                    stats["p1_trips"] += wn; stats["p1_lane"] += lane_tests.sum()
                stats["p2_trips"] += lane_pass.max(); stats["p2_lane"] += lane_pass.sum()
                p2round += lane_pass
            stats["p2_trips_round"] += p2round.max()
s = stats
print(s)
print("per tile-list entry per wave: p1 trips %.3f (quadrant-union %.3f)  p1 lane util %.2f | p2 trips %.3f lane util %.2f | p2 if whole-round window %.3f" % (
    s["p1_trips"]/s["entries"]/4, s["q_trips"]/s["entries"]/4, s["p1_lane"]/(64*s["p1_trips"]), s["p2_trips"]/s["entries"]/4, s["p2_lane"]/(64*s["p2_trips"]), s["p2_trips_round"]/s["entries"]/4))
