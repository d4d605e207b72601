"""Input hand-off: the DataLoaderX / BackgroundGenerator contract of data/utils/bg_dataloader.py."""
import time

import pytest
import torch
from torch.utils.data import Dataset

from exploremultimodal_amd.prefetch import BackgroundGenerator, DataLoaderX


class _DS(Dataset):
    def __len__(self):
        return 10

    def __getitem__(self, i):
        return {'image': torch.full((3, 4, 4), float(i)), 'text_ids': torch.arange(5) + i, 'meta': [torch.tensor(i)]}


def test_background_generator_order_bound_and_errors():
    produced = []

    def gen():
        for i in range(20):
            produced.append(i)
            yield i

    g = BackgroundGenerator(gen(), max_prefetch=3)
    time.sleep(0.2)
    assert len(produced) <= 5                  # bounded look-ahead: queue of 3 + one blocked put + one in hand
    assert list(g) == list(range(20))
    with pytest.raises(StopIteration):
        next(g)

    def bad():
        yield 1
        raise ValueError('boom')

    g = BackgroundGenerator(bad())
    assert next(g) == 1
    with pytest.raises(ValueError, match='boom'):
        next(g)


def test_dataloaderx_passthrough_without_gpu():
    dl = DataLoaderX(None, max_prefetch=2, dataset=_DS(), batch_size=4, shuffle=False)
    batches = list(dl)
    assert [b['image'].shape[0] for b in batches] == [4, 4, 2]
    assert torch.equal(batches[1]['text_ids'][0], torch.arange(5) + 4)
    assert len(list(dl)) == 3                  # re-iterable
    dl.shutdown()


@pytest.mark.gpu
def test_dataloaderx_uploads_on_a_side_stream():
    dl = DataLoaderX(0, max_prefetch=2, dataset=_DS(), batch_size=5, shuffle=False)
    seen = []
    for b in dl:
        assert b['image'].is_cuda and b['text_ids'].is_cuda and b['meta'][0].is_cuda
        seen.append(b['image'].sum().item())
    assert seen == [float(sum(range(5)) * 48), float(sum(range(5, 10)) * 48)]
    it = iter(dl)
    next(it)
    dl.shutdown()                              # stops the thread mid-epoch


def test_dataloaderx_attaches_the_head_row_indices():
    """The hand-off lists, on the host, the rows the MLM / MIM heads gather (objectives.attach_row_indices)."""
    from exploremultimodal_amd.prefetch import DataLoaderX

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(i)
            lab = torch.full((8,), -100, dtype=torch.int64)
            lab[torch.randperm(8, generator=g)[:2]] = 7
            bm = torch.zeros(2, 2, dtype=torch.int64)
            bm.view(-1)[i % 4] = 1
            return {'text_labels_mlm': lab, 'image_bool_masked_pos': bm, 'x': torch.zeros(3)}

    dl = DataLoaderX(None, batch_size=2, dataset=DS(), shuffle=False)
    seen = 0
    for b in dl:
        lab, bm = b['text_labels_mlm'], b['image_bool_masked_pos']
        assert torch.equal(b['_mlm_rows'], (lab.reshape(-1) != -100).nonzero().reshape(-1))
        flat = bm.reshape(bm.shape[0], -1) != 0
        assert torch.equal(b['_mim_rows'], flat.reshape(-1).nonzero().reshape(-1))
        full = torch.cat([torch.zeros(flat.shape[0], 1, dtype=torch.bool), flat], 1)
        assert torch.equal(b['_mim_tok_rows'], full.reshape(-1).nonzero().reshape(-1))
        seen += 1
    assert seen == 2
    dl.shutdown()
