import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.physics import *
n=4
ph=HectorPhysics(n)
rng=np.random.default_rng(0)
s=State(n)
s.root_pos[:,2]=5.0
s.q[:]=np.array([0,0,.785,-1.578,.785]*2)+rng.uniform(-.1,.1,(n,10))
s.qd[:]=rng.uniform(-2,2,(n,10))
s.root_angvel[:]=rng.uniform(-1,1,(n,3))
s.root_linvel[:]=rng.uniform(-1,1,(n,3))
z=np.zeros((n,10))
ke,pe=ph.energy(s); P0,L0=ph.momentum(s); E0=ke+pe
print('E0',E0)
# widen limits so joint limits do not act
ph.q_lo[:]=-100; ph.q_hi[:]=100; ph.v_max[:]=1e9
for k in range(300):
    ph.substep(s, z, z, z, z+100.0, dt=1e-4)
ke,pe=ph.energy(s); P1,L1=ph.momentum(s)
print('dE',ke+pe-E0)
print('dP',P1-P0 - np.array([0,0,GRAVITY])*ph.mass.sum(0)[:,None]*300*1e-4)
# angular momentum about COM conserved: L about origin changes by r_com x mg; check about-COM
print('dL (about origin, raw)',(L1-L0)[0])
