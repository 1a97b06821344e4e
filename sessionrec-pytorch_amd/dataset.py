"""Session files and prefix augmentation - host-side mirror of
/root/reference/src/utils/data/dataset.py (create_index :6-13, read_sessions :16-19,
read_dataset :22-27, AugmentedDataset :29-50).

On-disk format unchanged: train.txt / test.txt hold one session per line, comma-separated item
ids; num_items.txt one integer.  (The reference's pandas `squeeze=` call no longer exists in
pandas 2; plain parsing gives the same lists.)
"""
from pathlib import Path

import numpy as np


def create_index(sessions):
    """one sample per (session, label position >= 1), session-major: columns sessionId, labelIndex"""
    lens = np.fromiter((len(s) for s in sessions), dtype=np.int64, count=len(sessions))
    reps = np.maximum(lens - 1, 0)
    session_idx = np.repeat(np.arange(len(sessions)), reps)
    starts = np.cumsum(reps) - reps
    label_idx = np.arange(int(reps.sum())) - np.repeat(starts, reps) + 1
    return np.column_stack((session_idx, label_idx))


def read_sessions(filepath):
    out = []
    with open(filepath) as f:
        for line in f:
            line = line.strip()
            if line:
                out.append([int(x) for x in line.split(',')])
    arr = np.empty(len(out), dtype=object)
    arr[:] = out
    return arr


def read_dataset(dataset_dir):
    dataset_dir = Path(dataset_dir)
    train_sessions = read_sessions(dataset_dir / 'train.txt')
    test_sessions = read_sessions(dataset_dir / 'test.txt')
    with open(dataset_dir / 'num_items.txt', 'r') as f:
        num_items = int(f.readline())
    return train_sessions, test_sessions, num_items


class AugmentedDataset:
    def __init__(self, sessions, sort_by_length=False):
        self.sessions = sessions
        index = create_index(sessions)
        if sort_by_length:
            index = index[np.argsort(index[:, 1])[::-1]]
        self.index = index

    def __getitem__(self, idx):
        sid, lidx = self.index[idx]
        return self.sessions[sid][:lidx], self.sessions[sid][lidx]

    def __len__(self):
        return len(self.index)
