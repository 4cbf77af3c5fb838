"""Drop-in for the reference's ``dataloader`` module (dataloader.py:16-217) plus the device-resident
input pipeline of SURVEY.md 8f-1.

``DataSet`` keeps upstream's surface: it reads ``<BASE_PATH>/Data/model_data/split_processed/<cohort>.pickle``
(keys ``feat, padding_mask, time_step, drop_mask, encounter_id``), builds ``feed_data`` (N,4C,T) with the
value plane rescaled to +-scale/2, and ``__getitem__`` returns ``(sample, fake_sample)`` dicts, so a
``torch.utils.data.DataLoader`` over it behaves like the reference's.

``DeviceLoader`` is what the drivers use on a GPU: the whole cohort lives in HBM (75 k x 4*6*96 f32 =
0.7 GB), a batch is one gather, and the fake-sample corruption (dataloader.py:182-193) and the optional
Gaussian augmentation (dataloader.py:201-217) are drawn on the device for the whole batch at once instead
of per sample in NumPy worker processes.  It yields the same ``(batch_sample, fake_batch_sample)`` dict
pairs, with one extra key, ``lengths`` (B,C) int32, that lets the kernels skip the mask plane.
"""
import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

from .info import BASE_PATH, USE_FEATURES
from .utils import logger


class Transform(object):
    """float32 tensors + optional Gaussian noise on values / time stamps (dataloader.py:196-217)."""

    def __init__(self, aug, **aug_config):
        self.aug = aug
        self.ob_std = aug_config.get('ob_std', 0)

    def __call__(self, sample):
        for k, v in sample.items():
            if k != 'encounter_id':
                sample[k] = torch.as_tensor(v, dtype=torch.float32)
        if self.aug:
            sample['ob'] = self.add_gaussian_noise(sample['ob'], sample['padding_mask'], {'mean': 0., 'std': self.ob_std})
            sample['timestamp'] = self.add_gaussian_noise(sample['timestamp'], sample['padding_mask'], {'mean': 0., 'std': .01})
        return sample

    def add_gaussian_noise(self, tensor, padding_mask, gaussian_config_dict):
        mean, std = gaussian_config_dict.get('mean', 0.0), gaussian_config_dict.get('std', .1)
        return (tensor + torch.randn(tensor.size()) * std + mean) * padding_mask


class DataSet(Dataset):
    def __init__(self, args, cohort):
        self.use_features = USE_FEATURES
        self.num_features = getattr(args, 'num_variables', len(USE_FEATURES))
        self.logger = logger
        self.hours_from_admission = args.hours_from_admission
        self.cohort = cohort
        self.scale = args.scale
        self.aux_tasks = args.aux_tasks
        self.fake_detection = args.fake_detection
        self.data_path = os.path.join(BASE_PATH, 'Data')
        self.model_data_path = os.path.join(self.data_path, 'model_data')
        self.processed_data_path = os.path.join(self.model_data_path, 'split_processed')
        self.auxiliary_data_path = os.path.join(self.data_path, 'analysis_data')
        self.future_vital_path = os.path.join(self.data_path, 'vital_data')
        self._read_data()
        self._fix_input_format()
        self._scale_data(self.scale)
        if self.aux_tasks:
            self.auxiliary_dict = self._load_auxiliary_data()
        aug = bool(getattr(args, 'aug_input', False)) and cohort == 'training'
        self.aug_std = getattr(args, 'aug_std', 0.1)
        self.transform = Transform(aug=aug, **({'ob_std': self.aug_std} if aug else {}))

    # ---- loading (dataloader.py:49-79)
    def _read_data(self):
        self.data_f = os.path.join(self.processed_data_path, '{}.pickle'.format(self.cohort))
        with open(self.data_f, 'rb') as f:
            self.data_dict = pickle.load(f)

    def _fix_input_format(self):
        d = self.data_dict
        self.encounter_ids = d['encounter_id']
        planes = [d['feat'], d['padding_mask'], d['time_step'], d['drop_mask']]     # the order matters
        self.feed_data = np.concatenate([np.asarray(p, dtype=np.float64) for p in planes], axis=1)
        self.num_features = self.feed_data.shape[1] // 4
        self.num_timestep = self.feed_data.shape[-1]
        self.lengths = np.asarray(d['padding_mask']).sum(axis=-1).astype(np.int32)
        m = np.asarray(d['padding_mask'])
        self.prefix_masks = bool((m == (np.arange(m.shape[-1])[None, None] < self.lengths[..., None])).all())
        self.logger.info('{} data shape: {}'.format(self.cohort, self.feed_data.shape))

    def _scale_data(self, scale):
        if scale != 0:
            C = self.num_features
            self.feed_data[:, 0:C, :] = scale * self.feed_data[:, 0:C, :] - scale / 2
            self.logger.info('Scale input data to {}'.format(scale * np.array([0, 1]) - scale / 2))
        else:
            self.logger.info('No scale input, keep [0, 1]')

    def _load_auxiliary_data(self):
        """Supervised side labels from three CSVs of the private cohort (dataloader.py:81-113)."""
        import pandas as pd
        ids = pd.DataFrame(data={'encounter_deiden_id': self.encounter_ids})
        df = ids.merge(pd.read_csv(os.path.join(self.auxiliary_data_path, 'table_data.csv')))
        df = df.merge(pd.read_csv(os.path.join(self.auxiliary_data_path, 'mortality_summary.csv')))
        df = df.merge(pd.read_csv(os.path.join(self.future_vital_path, 'next_hour_abnormal_norm_val.csv')))
        out = {'encounter_deiden_id': ids['encounter_deiden_id'].values}
        if 'future_vital' in self.aux_tasks:
            fv = df[USE_FEATURES]
            out['future_vital_mask'] = fv.notnull().astype(int).to_numpy()
            out['future_vital'] = fv.fillna(0).to_numpy()
        for task in self.aux_tasks:
            if task == 'future_vital':
                continue
            if task not in ('AKI_overall', 'ICU_24h', 'ICU', 'mort_status_30d', 'mort_status_3y'):
                raise KeyError(task)
            out[task] = (df[task].values == 'Y').astype(int)
        return out

    # ---- per-sample access (dataloader.py:120-149)
    def __getitem__(self, index):
        C = self.num_features
        row = self.feed_data[index]
        eid = self.encounter_ids[index]
        ob, padding_mask, timestamp, ae_mask = row[0:C], row[C:2 * C], row[2 * C:3 * C], row[3 * C:]
        sample = {'encounter_id': eid, 'ob': ob, 'padding_mask': padding_mask, 'timestamp': timestamp, 'ae_mask': ae_mask}
        if self.fake_detection:
            fake_sample = {'encounter_id': eid, 'ob': self._generate_fake_data(ob, padding_mask),
                           'padding_mask': padding_mask, 'timestamp': timestamp, 'ae_mask': ae_mask}
        else:
            fake_sample = sample
        if self.aux_tasks:
            for task in self.aux_tasks:
                sample[task] = self.auxiliary_dict[task][index]
                if task == 'future_vital':
                    sample['future_vital_mask'] = self.auxiliary_dict['future_vital_mask'][index]
        # dataloader.py:147-148 transforms BOTH dicts -- without fake detection they are the same object, so the augmentation noise
        # is drawn and added twice on it (kept: it is what upstream's augmented pretraining sees)
        sample = self.transform(sample)
        fake_sample = self.transform(fake_sample)
        return sample, fake_sample

    def __len__(self):
        return len(self.feed_data)

    def _generate_fake_data(self, ob, padding_mask):
        """Replace half (at least one) of a channel's observed values by uniform noise (dataloader.py:182-193)."""
        fake = np.array(ob, copy=True)
        for values, mask in zip(fake, padding_mask):
            n_valid = int(np.sum(mask))
            n_perm = max(1, int(n_valid * 0.5))
            idx = np.random.choice(n_valid, size=n_perm, replace=False)
            noise = np.random.rand(n_perm)
            values[idx] = noise if self.scale == 0 else noise * self.scale - self.scale / 2
        return fake


def _same_device(a, b):
    """'cuda' and 'cuda:0' name the same card when 0 is the current device: compare with the index filled in."""
    a, b = torch.device(a), torch.device(b)
    if a.type != b.type:
        return False
    if a.type != 'cuda':
        return True
    cur = torch.cuda.current_device() if (a.index is None or b.index is None) else 0
    return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)


class DeviceLoader:
    """Iterates a ``DataSet`` held in HBM; yields ``(batch_sample, fake_batch_sample)`` dicts of device tensors.

    When the job is sharded (one process per GPU) each rank iterates its contiguous shard of every global batch."""

    def __init__(self, ds: DataSet, batch_size, shuffle, device, drop_last=False, seed=None, shard=True, keep_every_row=False,
                 ragged='auto', dense_samples=None, store=None):
        """``ragged`` ('auto' / True / False): keep the cohort as a ``ragged.RaggedStore`` (observed samples only, packed) instead of the
        padded (N,4C,T) array -- 'auto': whenever the cohort is prefix-masked with constant padding (what p0 writes).  Every sample dict
        then carries ``'ragged'``, a ``RaggedBatch`` the kernels read the store through.  ``dense_samples`` (None = automatic): also
        rebuild the padded per-batch tensors ('ob', 'padding_mask', 'timestamp', 'ae_mask'); automatic = only where something consumes
        them (unshuffled = evaluation / dump passes, fake-detection copies, augmentation).  ``store``: an existing ``RaggedStore`` of THIS
        dataset on this device (another loader's: the sharded evaluation loaders share the training loader's instead of packing and
        uploading the cohort a second time)."""
        from . import dist
        from .ragged import RaggedStore
        self.ds, self.batch_size, self.shuffle, self.device, self.drop_last = ds, int(batch_size), shuffle, device, drop_last
        self.C = ds.num_features
        if store is not None:
            if store.N != len(ds) or store.C != self.C or not _same_device(store.t_pk.device, device):
                raise ValueError('DeviceLoader(store=...): the store does not belong to this dataset / device')
            use_store = True
        else:
            use_store = ragged is True or (ragged == 'auto' and torch.device(device).type == 'cuda' and ds.prefix_masks
                                           and RaggedStore.fits(ds.feed_data, self.C))
            if ragged is True and not RaggedStore.fits(ds.feed_data, self.C):
                raise ValueError('DeviceLoader(ragged=True): the cohort is not prefix-masked with constant padding')
        self.store = store if store is not None else (RaggedStore(ds.feed_data, self.C, device) if use_store else None)
        self.data = None if use_store else torch.as_tensor(ds.feed_data, dtype=torch.float32, device=device)        # (N,4C,T)
        self.dense_samples = dense_samples
        self.lengths = torch.as_tensor(ds.lengths, device=device) if ds.prefix_masks else None
        self.ids = np.asarray(ds.encounter_ids)
        self.aux = {}
        if ds.aux_tasks:
            for k, v in ds.auxiliary_dict.items():
                if k != 'encounter_deiden_id':
                    self.aux[k] = torch.as_tensor(np.asarray(v), dtype=torch.float32, device=device)
        seed = int(seed if seed is not None else np.random.randint(0, 2 ** 31 - 1))
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.host_gen = torch.Generator()                # batch order is drawn on the host: ids need no D2H sync
        self.host_gen.manual_seed(seed)
        self.rank, self.world = (dist.rank(), dist.world_size()) if shard else (0, 1)
        self.dataset = ds
        # evaluation / feature passes must see every encounter: a trailing batch too small to shard is merged into the one before it
        self.keep_every_row = bool(keep_every_row)
        if min(len(ds), self.batch_size) < self.world:
            raise ValueError(f'DeviceLoader: batches of {min(len(ds), self.batch_size)} rows cannot be sharded over {self.world} ranks')

    def __len__(self):
        n = len(self.ds)
        full = n // self.batch_size
        tail = n - full * self.batch_size
        # a trailing batch with fewer rows than ranks would leave some rank an EMPTY shard (its kernels raise while the other ranks
        # wait in the all-reduces): it is dropped, identically on every rank
        keep_tail = tail >= self.world and not self.drop_last
        if self.keep_every_row and full == 0:
            return 1
        return full + (1 if keep_tail else 0)

    def _bounds(self, b, rank=None):
        """[lo, hi) of global batch b in the epoch's order, and this rank's (or ``rank``'s) [lo, hi) inside it."""
        n, nb = len(self.ds), len(self)
        rank = self.rank if rank is None else rank
        lo = b * self.batch_size
        hi = n if (b == nb - 1 and self.keep_every_row) else min((b + 1) * self.batch_size, n)
        if self.world > 1:
            m = hi - lo
            return lo, hi, lo + (m * rank) // self.world, lo + (m * (rank + 1)) // self.world
        return lo, hi, lo, hi

    def shard_rows(self, rank=None):
        """Dataset rows this rank (or ``rank``) yields over one UNSHUFFLED pass, in iteration order (the scatter index of trainers' row gather)."""
        if self.shuffle:
            raise ValueError('shard_rows: the loader shuffles')
        parts = [torch.arange(*self._bounds(b, rank)[2:]) for b in range(len(self))]
        return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)

    def _noise(self, t, mask, std):
        return (t + torch.randn(t.shape, device=self.device, generator=self.gen) * std) * mask

    def _fake_ob(self, ob, mask, lengths):
        """Batched form of DataSet._generate_fake_data: per (b,c) row, max(1, n//2) distinct valid slots."""
        n = lengths if lengths is not None else mask.sum(-1).to(torch.int32)
        key = torch.rand(ob.shape, device=self.device, generator=self.gen)
        key = torch.where(mask > 0, key, torch.full_like(key, 2.0))
        rank = key.argsort(dim=-1).argsort(dim=-1)                       # rank of each slot among the row's random keys
        n_perm = torch.clamp((n.to(torch.float32) * 0.5).floor().to(torch.int64), min=1)
        hit = (rank < n_perm[..., None]) & (mask > 0)
        noise = torch.rand(ob.shape, device=self.device, generator=self.gen)
        if self.ds.scale != 0:
            noise = noise * self.ds.scale - self.ds.scale / 2
        return torch.where(hit, noise, ob)

    def __iter__(self):
        n = len(self.ds)
        # every rank draws the same permutation (same seed), then takes its shard of each batch
        order_h = torch.randperm(n, generator=self.host_gen) if self.shuffle else torch.arange(n)
        order = order_h.to(self.device)
        C = self.C
        for b in range(len(self)):
            glo, ghi, lo, hi = self._bounds(b)
            idx, idx_h = order[lo:hi], order_h[lo:hi].numpy()
            lengths = None if self.lengths is None else self.lengths.index_select(0, idx)
            aug = self.ds.transform.aug
            rb = None
            if self.store is not None:
                from .ragged import RaggedBatch
                rb = RaggedBatch(self.store, idx, lengths)
                dense = self.dense_samples if self.dense_samples is not None else (not self.shuffle or self.ds.fake_detection or aug)
                if not dense:          # the training pass of the plain objectives: nothing reads the padded tensors
                    sample = {'encounter_id': self.ids[idx_h], 'lengths': lengths, 'ragged': rb, 'global_rows': ghi - glo}
                    for k, v in self.aux.items():
                        sample[k] = v.index_select(0, idx)
                    yield sample, sample
                    continue
                rows = self.store.dense_rows(idx)
            else:
                rows = self.data.index_select(0, idx)
            ob, mask, ts, ae = rows[:, 0:C], rows[:, C:2 * C], rows[:, 2 * C:3 * C], rows[:, 3 * C:4 * C]
            raw_ob, raw_ts = ob, ts
            # the per-sample Transform of DataSet.__getitem__, batched: the real and the fake sample get independent noise; without
            # fake detection both names are ONE dict upstream, which therefore receives the noise twice (dataloader.py:147-148)
            for _ in range(0 if not aug else (1 if self.ds.fake_detection else 2)):
                ob = self._noise(ob, mask, self.ds.aug_std)
                ts = self._noise(ts, mask, 0.01)
            sample = {'encounter_id': self.ids[idx_h], 'ob': ob, 'padding_mask': mask, 'timestamp': ts,
                      'ae_mask': ae, 'lengths': lengths, 'global_rows': ghi - glo}     # (global_rows: rows of the GLOBAL batch -- step.Stepper's rank-invariant graph key)
            for k, v in self.aux.items():
                sample[k] = v.index_select(0, idx)
            if rb is not None and not aug:
                sample['ragged'] = rb             # (augmented values exist as padded tensors only)
            if self.ds.fake_detection:
                fake = dict(sample)
                fake.pop('ragged', None)
                fake['ob'] = self._fake_ob(raw_ob, mask, lengths)         # corrupted copy of the un-augmented values (dataloader.py:131-132)
                fake['timestamp'] = raw_ts
                if aug:
                    fake['ob'] = self._noise(fake['ob'], mask, self.ds.aug_std)
                    fake['timestamp'] = self._noise(raw_ts, mask, 0.01)
            else:
                fake = sample
            yield sample, fake
