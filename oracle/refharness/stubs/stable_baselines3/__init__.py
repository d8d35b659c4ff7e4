"""Name-only stand-in so that the reference's src/rl_utils.py:15-17 imports resolve (oracle harness only).
Nothing here is ever called: the harness uses only load_data / Preprocessing from that module."""
