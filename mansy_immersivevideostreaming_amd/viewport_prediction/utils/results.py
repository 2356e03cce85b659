"""Host mirror of viewport_prediction/utils/results.py: `Results.record/write/reset` (:53-152) with the per-sample
metrics (periodic MSE, tile IoU "accuracy", recall, precision, F1) computed in batch on the device; the CSV / log /
per-horizon accuracy files keep the reference's formats."""
import os
from collections import namedtuple

import numpy as np

from .common import compute_accuracy, get_config_from_yml, mean_square_error

Key = namedtuple('Key', 'video user timestamp')
Value = namedtuple('Value', 't gt pred mse accuracy prob recall precision f1')


class Results:
    def __init__(self, model_name, dimension, fut_window, output_dir, dataset_frequency, mse=True, nll=False, accuracy=False, config=None):
        self.config = config if config is not None else get_config_from_yml()
        self.model_name, self.dimension, self.fut_window, self.output_dir = model_name, dimension, fut_window, output_dir
        self.mse, self.nll, self.accuracy = mse, nll, accuracy
        self.results = []
        self.dataset_frequency = dataset_frequency
        self.accuracy_results = [[] for _ in range(fut_window)]

    def record(self, batch_size, prediction, ground_truth, video, user, timestamp):
        mse_arr = mean_square_error(prediction, ground_truth).cpu().numpy() if self.mse else None
        if self.accuracy:
            acc, rec, prec, f1 = compute_accuracy(ground_truth, prediction, self.config.video_width, self.config.video_height,
                                                  self.config.tile_num_width, self.config.tile_num_width)
        pred_h, gt_h = prediction.cpu().numpy(), ground_truth.cpu().numpy()
        for i in range(batch_size):
            key = Key(video=video[i], user=int(user[i]), timestamp=int(timestamp[i]))
            values = []
            for t in range(self.fut_window):
                mse = float(mse_arr[i, t]) if self.mse else None
                accuracy = recall = precision = f1v = None
                if self.accuracy:
                    accuracy, recall, precision, f1v = acc[i, t], rec[i, t], prec[i, t], f1[i, t]
                    self.accuracy_results[t].append(accuracy)
                values.append(Value(t=round((t + 1) * (1 / self.dataset_frequency), 3), gt=gt_h[i][t], pred=pred_h[i][t], mse=mse,
                                    accuracy=accuracy, prob=None, recall=recall, precision=precision, f1=f1v))
            self.results.append({key: values})

    def write(self, log=True, label=''):
        csv_path = os.path.join(self.output_dir, label + 'results.csv')
        with open(csv_path, 'w', encoding='utf-8') as csv_file:
            csv_file.write('video,user,timestamp,time,gt_1,gt_2,pred_1,pred_2,mse,accuracy,recall,precision,f1\n')
            for element in self.results:
                for key, values in element.items():
                    for value in values:
                        line = f'{key.video},{key.user},{key.timestamp},{value.t},{value.gt[0]},{value.gt[1]},'
                        for i in range(len(value.pred)):
                            line += f'{value.pred[i]},'
                        line += f'{value.mse},{value.accuracy},{value.recall},{value.precision},{value.f1}\n'
                        csv_file.write(line)
        print('Results saved at', csv_path)
        if log:
            log_path = os.path.join(self.output_dir, label + 'results.log')
            with open(log_path, 'w', encoding='utf-8') as f:
                for element in self.results:
                    for key, values in element.items():
                        f.write(f'##### Video={key[0]}, User={key[1]}, Timestamp={key[2]} #####\n')
                        for value in values:
                            f.write(f'time={value[0]}, gt={list(value[1])}, pred={list(value[2])}, mse={value[3]}, accuracy={value[5]}, '
                                    f'recall={value[5]}, precision={value[6]}, f1={value[7]}\n')
            print('Log saved at', log_path)
        if self.accuracy:
            accuracy_csv_path = os.path.join(self.output_dir, label + 'accuracy_result.csv')
            mean_accuracy = []
            with open(accuracy_csv_path, 'w', encoding='utf-8') as csv_file:
                csv_file.write('timestamp,accuracy\n')
                for i in range(self.fut_window):
                    mean_accuracy.append(sum(self.accuracy_results[i]) / len(self.accuracy_results[i]) * 100.)
                    csv_file.write(f'{round((i + 1) * (1 / self.dataset_frequency), 3)},{mean_accuracy[i]}\n')
            running = [sum(mean_accuracy[:i + 1]) / (i + 1) for i in range(self.fut_window)]
            print('Mean accuracy up to each horizon:')
            print(' | '.join(f'{round((i + 1) * (1 / self.dataset_frequency), 3)}s: {round(m, 5)}' for i, m in enumerate(running)))
            return running

    def reset(self):
        self.results.clear()
        self.accuracy_results = [[] for _ in range(self.fut_window)]
