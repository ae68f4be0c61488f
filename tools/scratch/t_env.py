import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.env import *
for name in ['env_rollout_a','env_rollout_b']:
    fx=np.load(f'/root/repo/tests/golden/{name}.npz')
    n,steps,seed,sc0,noise=fx['meta']
    env=HectorEnvOracle(n, fx['init_shape_friction'], fx['init_base_mass'], fx['init_env_origins'], fx['packs'][0], add_noise=bool(noise), start_xy=fx['init_start_pos'])
    print(name,'init obs err', np.abs(env.obs_buf-fx['init_obs_full']).max(), 'priv', np.abs(env.priv_buf-fx['init_priv_full']).max())
    env.episode_length_buf[:]=fx['ep_len_init']; env.common_step_counter=int(sc0)
    worst={}
    for t in range(steps):
        obs,priv,rew,reset=env.step(fx['actions'][t], fx['packs'][t+1])
        errs=dict(obs=np.abs(obs[:,-41:]-fx['obs41'][t]).max(), priv=np.abs(priv[:,-70:]-fx['priv70'][t]).max(),
                  rew=np.abs(rew-fx['rew'][t]).max(), reset=np.abs(reset.astype(int)-fx['reset'][t]).max(),
                  to=np.abs(env.time_out_buf.astype(int)-fx['timeout'][t]).max(),
                  tov=np.abs(env.time_outs_visible.astype(int)-fx['timeouts_visible'][t]).max(),
                  q=np.abs(env.state.q-fx['q'][t]).max(), tau=np.abs(env.torques-fx['torques'][t]).max(),
                  cmd=np.abs(env.commands-fx['commands'][t]).max(), fat=np.abs(env.feet_air_time-fx['feet_air_time'][t]).max(),
                  fh=np.abs(env.feet_height-fx['feet_height'][t]).max(),
                  es=np.abs(np.stack([env.episode_sums[k] for k in REWARD_ORDER])-fx['episode_sums'][t]).max())
        for k,v in errs.items(): worst[k]=max(worst.get(k,0),v)
        if t+1 in fx['full_steps']:
            i=list(fx['full_steps']).index(t+1)
            worst['full_obs']=max(worst.get('full_obs',0),np.abs(obs-fx['full_obs'][i]).max())
            worst['full_priv']=max(worst.get('full_priv',0),np.abs(priv-fx['full_priv'][i]).max())
    print(name, {k:float('%.3g'%v) for k,v in worst.items()})
    print(list(fx['reward_names'])==REWARD_ORDER, np.abs(fx['reward_scales']-np.array([REWARD_SCALE[k]*0.01 for k in REWARD_ORDER])).max())
