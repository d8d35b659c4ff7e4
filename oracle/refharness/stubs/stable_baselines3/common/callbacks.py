EvalCallback = None
