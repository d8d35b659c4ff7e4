"""TEST INFRASTRUCTURE -- CPU restatement of SB3's reward normalisation, used only by tests/ as the checker.

stable-baselines3 == 2.0.0a13 is the reference's pin (/root/reference/requirements.txt:5) and is NOT vendored in the reference
nor installable here, so this file restates its published algorithm; the reference's call site is
`VecNormalize(env, norm_obs=False)` (src/rl_utils.py:453), i.e. defaults training=True, norm_reward=True, clip_reward=10.0,
gamma=0.99, epsilon=1e-8.  Parity against SB3 itself is therefore UNPINNED (no golden vectors of it exist in the reference).

Restated from stable_baselines3/common/running_mean_std.py (RunningMeanStd.__init__/update/update_from_moments) and
stable_baselines3/common/vec_env/vec_normalize.py (VecNormalize.step_wait, _update_reward, normalize_reward).
"""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, arr):
        batch_mean = np.mean(arr, axis=0)
        batch_var = np.var(arr, axis=0)
        batch_count = arr.shape[0]
        self.update_from_moments(batch_mean, batch_var, batch_count)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        m_2 = m_a + m_b + np.square(delta) * self.count * batch_count / (self.count + batch_count)
        new_var = m_2 / (self.count + batch_count)
        new_count = batch_count + self.count
        self.mean = new_mean
        self.var = new_var
        self.count = new_count


class RewardNormalizer:
    """The reward half of VecNormalize (norm_obs=False): step(rewards [N], dones [N]) -> normalised rewards [N]."""

    def __init__(self, num_envs, gamma=0.99, epsilon=1e-8, clip_reward=10.0, training=True):
        self.ret_rms = RunningMeanStd(shape=())
        self.returns = np.zeros(num_envs)
        self.gamma, self.epsilon, self.clip_reward, self.training = gamma, epsilon, clip_reward, training

    def step(self, rewards, dones):
        if self.training:                                   # _update_reward
            self.returns = self.returns * self.gamma + rewards
            self.ret_rms.update(self.returns)
        out = np.clip(rewards / np.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward, self.clip_reward)
        self.returns[np.asarray(dones, dtype=bool)] = 0
        return out

    def rollout(self, rewards, dones):                      # [T, N] -> [T, N]
        return np.stack([self.step(rewards[t], dones[t]) for t in range(len(rewards))])
