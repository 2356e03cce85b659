"""Host mirror of the helper pieces of bitrate_selection/utils/common.py used by the drivers: config loading (:13-37),
read_log_file (:196-218).  (normalisers, action2rates and allocate_tile_rates live inside csrc/env.hip.)"""
from ...viewport_prediction.utils.common import Config, get_config_from_yml  # noqa: F401  (identical loader, common.py:13-37)


def read_log_file(log_path, verbose=True):
    rows, means = [], [0.0, 0.0, 0.0, 0.0]
    with open(log_path, 'r') as file:
        file.readline()
        for line in file.readlines():
            line = line.strip().split(',')
            video, user, trace = list(map(int, line[:3]))
            vals = list(map(float, line[3:]))
            for i in range(4):
                means[i] += vals[3 + i]
            rows.append([video, user, trace] + vals)
    n = max(len(rows), 1)
    means = [m / n for m in means]
    if verbose:
        print('video user trace qoe_w1 qoe_w2 qoe_w3 qoe qoe1 qoe2 qoe3')
        for r in rows:
            print(' '.join(str(x) for x in r))
        print(-1, -1, -1, -1, -1, -1, *means)
    return means
