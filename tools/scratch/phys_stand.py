import sys; sys.path.insert(0,'/root/repo')
import numpy as np, time
from oracle.physics import *
n=3
ph=HectorPhysics(n, base_mass_added=[0,-2,4], shape_friction=[1.0,0.1,0.5])
s=State(n)
s.root_pos[:,2]=0.55
q0=np.array([0,0,.785,-1.578,.785]*2)
s.q[:]=q0
kp=np.array([40,40,60,120,20]*2,float); kd=np.array([3,3,5,4,1]*2,float)
tl=0.85*np.array([33.5,33.5,33.5,67,33.5]*2)
bs=ph.body_states(s)
print('toe z',bs[:, [5,10], 2], 'calf z', bs[:,[4,9],2])
t=time.time()
for k in range(2000):
    ph.substep(s, np.broadcast_to(q0,(n,10)), kp, kd, tl)
    if k%250==0 or k==1999:
        print(k, 'z',np.round(s.root_pos[:,2],4), 'Fz feet', np.round(ph.contact_force[:,[5,10],2],1), 'sumF', np.round(ph.contact_force[:,:,2].sum(1),2), 'mg', np.round(ph.mass.sum(0)*9.81,2))
print('time',time.time()-t)
print('pos',s.root_pos, 'quat', s.root_quat)
print('q-q0', np.round(s.q-q0,3))
print('tau', np.round(ph.tau,2))
