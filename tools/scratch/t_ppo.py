import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.ppo import *
from tests.ppo_inputs import rollout_inputs
for name in ['ppo_small','ppo_clip']:
    fx=np.load(f'/root/repo/tests/golden/{name}.npz')
    seed,T,N,ep,nmb=fx['meta']
    ac=ActorCriticOracle.default_init(np.random.default_rng(seed))
    init={k:v.copy() for k,v in ac.state_dict().items()}
    alg=PPOOracle(ac,N,T,learning_rate=float(fx['lr0']))
    inp=rollout_inputs(seed,T,N)
    e={}
    for t in range(T):
        a=alg.act(inp['obs'][t],inp['priv'][t],inp['eps'][t])
        e['act']=max(e.get('act',0),np.abs(a-fx['actions'][t]).max())
        e['val']=max(e.get('val',0),np.abs(alg._tr['v']-fx['values'][t]).max())
        e['logp']=max(e.get('logp',0),np.abs(alg._tr['logp']-fx['logp'][t]).max())
        alg.process_env_step(inp['rewards'][t]*np.float32(fx['scale_rewards']),inp['dones'][t],inp['time_outs'][t])
    alg.compute_returns(inp['priv'][T])
    e['rew']=np.abs(alg.rewards-fx['stored_rewards']).max()
    e['ret']=np.abs(alg.returns-fx['returns']).max()
    e['adv']=np.abs(alg.advantages-fx['advantages']).max()
    mvl,msl=alg.update(fx['perm'])
    print(name,{k:float('%.3g'%v) for k,v in e.items()})
    print(' vloss',mvl,float(fx['mean_value_loss']),'sloss',msl,float(fx['mean_surrogate_loss']))
    print(' lr',np.array(alg.lr_hist)/fx['lrs'], 'gn rel', np.array(alg.gnorm_hist)/fx['grad_norms']-1)
    sd=ac.state_dict()
    for k in sd:
        d=sd[k].astype(np.float64)-init[k]
        print('  %-16s dsum %.4e ref %.4e | dabs %.6e ref %.6e | slice err %.2e'%(k,d.sum(),fx['delta_sum_'+k],np.abs(d).sum(),fx['delta_abs_'+k],np.abs(sd[k].reshape(-1)[:64]-fx['slice_'+k]).max()))
    print(' adam m std err',np.abs(alg.m[0]-fx['adam_m_std']).max(), 'v', np.abs(alg.v[0]-fx['adam_v_std']).max(), np.abs(alg.m[1].reshape(-1)[:64]-fx['adam_m_actor0_slice']).max())
