"""Loading of the reference's checkpoints (.pth written by mmcv's save_checkpoint:
{'meta':..., 'state_dict':...}; BasicVSR checkpoints prefix generator weights with 'generator.',
DDP-wrapped ones with 'module.').  Mirrors mmcv.runner.load_checkpoint as used at
mmedit/models/backbones/sr_backbones/iconvsr.py:510-523 and tools/test.py:129,156-159."""
import torch


def _strip(sd, prefix):
    if sd and all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return sd


def load_state_dict_file(filename, map_location='cpu'):
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        ckpt = ckpt['state_dict']
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    return _strip(dict(ckpt), 'module.')


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    sd = load_state_dict_file(filename, map_location)
    own = set(model.state_dict().keys())
    if not (set(sd) & own):
        # a generator loaded from a BasicVSR checkpoint, or the other way round
        gen = {k[len('generator.'):]: v for k, v in sd.items() if k.startswith('generator.')}
        if set(gen) & own:
            sd = gen
    missing, unexpected = model.load_state_dict(sd, strict=False)
    msg = []
    if unexpected:
        msg.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msg.append('missing keys in source state_dict: ' + ', '.join(missing))
    if msg:
        if strict:
            raise RuntimeError('\n'.join(msg))
        if logger is not None:
            logger.warning('\n'.join(msg))
    return sd
