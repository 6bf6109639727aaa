"""Token vocabulary / edge-type constants of the Polyphemus data format.

Restates the values the hot path depends on (reference `constants.py:4,22-58`);
they are part of the input contract, not tunables.
"""

N_TRACKS = 4                 # constants.py:4  (track 0 = drums)
N_TIMESTEPS = 32             # resolution 8 * 4 timesteps per bar (model.py:437-438)

PITCH_SOS, PITCH_EOS, PITCH_PAD = 128, 129, 130     # constants.py:22-25
N_PITCH_TOKENS = 131                                # constants.py:28
MAX_PITCH_TOKEN = 127

DUR_SOS, DUR_EOS, DUR_PAD = 96, 97, 98              # constants.py:35-38
N_DUR_TOKENS = 99                                   # constants.py:40
MAX_DUR_TOKEN = 95

D_TOKEN_PAIR = N_PITCH_TOKENS + N_DUR_TOKENS        # 230, constants.py:45
MAX_SIMU_TOKENS = 16                                # constants.py:48 (SOS + 14 notes + EOS)
N_SLOTS = MAX_SIMU_TOKENS - 1                       # 15 (SOS dropped, model.py:349)

EDGE_TRACK, EDGE_ONSET, EDGE_NEXT = 0, N_TRACKS, N_TRACKS + 1   # constants.py:52-55
N_EDGE_TYPES = N_TRACKS + 2                         # 6, constants.py:58
N_DISTS = 32                                        # one-hot timestep distance width (data.py:179)
