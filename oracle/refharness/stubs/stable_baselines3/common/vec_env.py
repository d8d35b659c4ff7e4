VecNormalize = DummyVecEnv = SubprocVecEnv = None
