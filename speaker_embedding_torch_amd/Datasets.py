"""Host-side batch assembly over the reference's on-disk pattern format (behaviour of reference Datasets.py:9-109).

Pattern file: pickle {'Mel': float16[Mel_dim, T_full], 'Speaker': str, 'Dataset': str} as written by the
reference's Pattern_Generator.py:191-198; METADATA.PICKLE carries 'File_List_by_Speaker_Dict'.  Only files
that this project's users generated themselves are unpickled.  A training batch is float32
[Speakers * Pattern_per_Speaker, Mel, T], speaker-major, with ONE random T per batch; an inference batch
stacks `samples` overlapping windows per utterance.  The mel stays fp16 on disk and is widened once here -- or,
with `half=True`, not at all: the batch then stays float16 through pinned memory and PCIe (half the
bytes) and the HIP path's packing kernel widens it on the device, with identical results.  `DevicePrefetcher` copies
batch i+1 on its own stream while step i runs.
"""
import os
import pickle
import random

import numpy as np
import torch


def Correction(feature, frame_length):
    """Bring a [Mel, T_full] pattern to exactly frame_length frames: random crop when longer, symmetric
    reflect padding (floor/ceil split) when shorter -- Datasets.py:9-19."""
    surplus = feature.shape[1] - frame_length
    if surplus > 0:
        start = np.random.randint(0, surplus)
        return feature[:, start:start + frame_length]
    left = (-surplus) // 2
    return np.pad(feature, ((0, 0), (left, -surplus - left)), mode="reflect")


def _load_pickle(path):
    with open(path.replace("\\", "/"), "rb") as handle:
        return pickle.load(handle)


def _to_batch(list_of_arrays, half=False):
    return torch.from_numpy(np.ascontiguousarray(np.stack(list_of_arrays, axis=0), dtype=np.float16 if half else np.float32))


class Dataset(torch.utils.data.Dataset):
    """Index = speaker; item = `pattern_per_speaker` randomly drawn (mel, speaker) pairs (Datasets.py:22-69).
    Speakers with fewer files than that are dropped; `num_speakers` optionally subsamples the speaker set."""

    def __init__(self, pattern_path, metadata_file, pattern_per_speaker, num_speakers=None):
        self.pattern_path = pattern_path
        self.pattern_per_speaker = pattern_per_speaker
        table = _load_pickle(os.path.join(pattern_path, metadata_file))["File_List_by_Speaker_Dict"]
        usable = {spk: files for spk, files in table.items() if len(files) >= pattern_per_speaker}
        if num_speakers is not None and num_speakers < len(usable):
            usable = {spk: usable[spk] for spk in random.sample(list(usable), num_speakers)}
        self.files_by_speakers = usable
        self.speakers = list(usable)

    def __len__(self):
        return len(self.speakers)

    def __getitem__(self, idx):
        speaker = self.speakers[idx]
        chosen = random.sample(self.files_by_speakers[speaker], self.pattern_per_speaker)
        return [(_load_pickle(os.path.join(self.pattern_path, name))["Mel"], speaker) for name in chosen]


class Collater:
    """Train/eval collate_fn (Datasets.py:72-86)."""

    def __init__(self, min_frame_length, max_frame_length, half=False):
        self.min_frame_length, self.max_frame_length, self.half = min_frame_length, max_frame_length, half

    def __call__(self, batch):
        frames = np.random.randint(self.min_frame_length, self.max_frame_length + 1)
        return _to_batch([Correction(mel, frames) for item in batch for mel, _ in item], self.half)


class Inference_Collater:
    """`samples` windows of `frame_length` with hop frame_length - overlap_length per utterance, stacked to
    [Speakers * Samples, Mel_dim, Time]; returns (features, speaker labels) -- Datasets.py:88-109."""

    def __init__(self, samples, frame_length, overlap_length, half=False):
        self.samples, self.frame_length, self.overlap_length, self.half = samples, frame_length, overlap_length, half
        self.required_length = samples * (frame_length - overlap_length) + overlap_length

    def slices(self, feature):
        feature = Correction(feature, self.required_length)
        hop = self.frame_length - self.overlap_length
        return [feature[:, s:s + self.frame_length] for s in range(0, self.required_length - self.overlap_length, hop)]

    def __call__(self, batch):
        windows, speakers = [], []
        for item in batch:
            for mel, speaker in item:
                windows.extend(self.slices(mel))
                speakers.append(speaker)
        return _to_batch(windows, self.half), speakers


class DevicePrefetcher:
    """Iterate a DataLoader one batch ahead on the device: batch i+1 is copied (pinned memory -> HBM, on a stream of
    its own) while the caller trains on batch i, so the step never waits for PCIe.  Items may be tensors or tuples /
    lists whose tensor members are moved (Inference_Collater returns (features, labels))."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _move(self, item):
        if torch.is_tensor(item):
            return item.to(self.device, non_blocking=True)
        if isinstance(item, (tuple, list)):
            return type(item)(self._move(x) for x in item)
        return item

    @staticmethod
    def _record(item, stream):
        if torch.is_tensor(item):
            if item.is_cuda:
                item.record_stream(stream)
        elif isinstance(item, (tuple, list)):
            for x in item:
                DevicePrefetcher._record(x, stream)

    def __iter__(self):
        if self.stream is None:
            yield from self.loader
            return
        pending = None
        for item in self.loader:
            with torch.cuda.stream(self.stream):
                moved = self._move(item)
            ready = torch.cuda.Event()
            ready.record(self.stream)
            if pending is not None:
                yield self._hand_over(*pending)
            pending = (moved, ready)
        if pending is not None:
            yield self._hand_over(*pending)

    def _hand_over(self, moved, ready):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        self._record(moved, cur)
        return moved
