make_vec_env = None
