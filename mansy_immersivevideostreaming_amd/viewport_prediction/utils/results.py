"""Evaluation notebook of the VP test loop -- counterpart of `Results` (reference: viewport_prediction/utils/results.py
:53-152: record / write / reset) with a columnar design: every `record()` call appends whole device-computed metric
blocks ([B, T] arrays of periodic MSE, tile IoU "accuracy", recall, precision, F1 from the HIP kernels); `write()` streams
them out in the reference's three file formats (`*results.csv`, `*results.log`, `*accuracy_result.csv`)."""
import os

import numpy as np

from .common import compute_accuracy, get_config_from_yml, mean_square_error

_CSV_COLUMNS = 'video,user,timestamp,time,gt_1,gt_2,pred_1,pred_2,mse,accuracy,recall,precision,f1'


class Results:
    def __init__(self, model_name, dimension, fut_window, output_dir, dataset_frequency, mse=True, nll=False, accuracy=False, config=None):
        self.config = get_config_from_yml() if config is None else config
        self.model_name, self.dimension, self.fut_window = model_name, dimension, fut_window
        self.output_dir, self.dataset_frequency = output_dir, dataset_frequency
        self.mse, self.nll, self.accuracy = mse, nll, accuracy
        self.reset()

    def reset(self):
        self._blocks = []          # one dict of arrays per recorded batch

    @property
    def horizons(self):
        return [round((t + 1) * (1 / self.dataset_frequency), 3) for t in range(self.fut_window)]

    def record(self, batch_size, prediction, ground_truth, video, user, timestamp):
        blk = {'video': np.asarray(video)[:batch_size], 'user': np.asarray(user)[:batch_size].astype(np.int64),
               'timestamp': np.asarray(timestamp)[:batch_size].astype(np.int64),
               'gt': ground_truth[:batch_size].cpu().numpy(), 'pred': prediction[:batch_size].cpu().numpy()}
        if self.mse:
            blk['mse'] = mean_square_error(prediction[:batch_size], ground_truth[:batch_size]).cpu().numpy()
        if self.accuracy:
            c = self.config
            # (the reference passes tile_num_width for both grid dimensions, results.py:73-75)
            blk['accuracy'], blk['recall'], blk['precision'], blk['f1'] = compute_accuracy(
                ground_truth[:batch_size], prediction[:batch_size], c.video_width, c.video_height, c.tile_num_width, c.tile_num_width)
        self._blocks.append(blk)

    def _rows(self):
        for blk in self._blocks:
            for i in range(len(blk['user'])):
                yield blk, i

    def _metric(self, blk, name, i, t):
        if name not in blk:
            return None
        return float(blk[name][i, t]) if name == 'mse' else blk[name][i, t]      # the reference stores mse as `.item()` (results.py:81)

    def write(self, log=True, label=''):
        stem = os.path.join(self.output_dir, label)
        hz = self.horizons
        with open(stem + 'results.csv', 'w', encoding='utf-8') as out:
            out.write(_CSV_COLUMNS + '\n')
            for blk, i in self._rows():
                head = f"{blk['video'][i]},{blk['user'][i]},{blk['timestamp'][i]},"
                for t in range(self.fut_window):
                    g, p = blk['gt'][i, t], blk['pred'][i, t]
                    cells = [hz[t], g[0], g[1], *p, *(self._metric(blk, k, i, t) for k in ('mse', 'accuracy', 'recall', 'precision', 'f1'))]
                    out.write(head + ','.join(f'{c}' for c in cells) + '\n')          # f-string formatting like results.py:103-110
        print('Results saved at', stem + 'results.csv')
        if log:
            with open(stem + 'results.log', 'w', encoding='utf-8') as out:
                for blk, i in self._rows():
                    out.write(f"##### Video={blk['video'][i]}, User={blk['user'][i]}, Timestamp={blk['timestamp'][i]} #####\n")
                    for t in range(self.fut_window):
                        m = {k: self._metric(blk, k, i, t) for k in ('mse', 'accuracy', 'recall', 'precision', 'f1')}
                        # The reference indexes its value tuple one slot late (results.py:120-122): `accuracy` and `recall`
                        # print the always-None `prob` field, `precision` prints recall and `f1` prints precision.  Kept
                        # byte-compatible so existing log parsers see the same file.
                        out.write(f"time={hz[t]}, gt={list(blk['gt'][i, t])}, pred={list(blk['pred'][i, t])}, mse={m['mse']}, "
                                  f"accuracy=None, recall=None, precision={m['recall']}, f1={m['precision']}\n")
            print('Log saved at', stem + 'results.log')
        if not self.accuracy or not self._blocks:
            return None
        acc = np.concatenate([blk['accuracy'] for blk in self._blocks], axis=0)             # [n, T]
        per_horizon = [sum(acc[:, t].tolist()) / acc.shape[0] * 100. for t in range(self.fut_window)]   # python-float sums like the reference
        with open(stem + 'accuracy_result.csv', 'w', encoding='utf-8') as out:
            out.write('timestamp,accuracy\n')
            for h, a in zip(hz, per_horizon):
                out.write(f'{h},{a}\n')
        running = [sum(per_horizon[:t + 1]) / (t + 1) for t in range(self.fut_window)]
        print('mean accuracy up to each horizon: ' + ' | '.join(f'{h}s {round(a, 5)}' for h, a in zip(hz, running)))
        return running
