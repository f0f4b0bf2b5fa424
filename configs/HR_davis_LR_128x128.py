# Test-path configuration with the reference's keys (configs/HR_davis_LR_128x128.py:4-29,134-206,238 of
# ZeldaM1/PnP-VCVE).  model / test_cfg / dist_params are key-for-key what the reference builds; the
# dataset entry points at the synthetic clip source of this build because the REDS/DAVIS files and the
# on-disk loader are outside the hot path.
exp_name = 'HR_davis_LR_128x128'

model = dict(
    type='BasicVSR',
    generator=dict(
        type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par',
        mid_channels=64, num_blocks=8, padding=3, with_cat=True, use_base_qp=True, num_experts=6,
        expert_softmax=True, init_weight=True, with_bias=True, with_se=True, with_par=True, one_layer=True,
        blocktype='drt', channel_first=True, sparse_val=False, align_key=True, vsr=False),
    pixel_loss=dict(type='CharbonnierLoss', loss_weight=1.0, reduction='mean'))
train_cfg = dict(fix_iter=5000)
test_cfg = dict(metrics=['PSNR', 'SSIM'], crop_border=0)

val_dataset_type = 'SyntheticCompressedClipDataset'
data = dict(
    workers_per_gpu=6,
    test_dataloader=dict(samples_per_gpu=1, workers_per_gpu=1),
    test=dict(type=val_dataset_type, num_clips=8, num_input_frames=7, height=128, width=128, slices='IBBBP',
              qp_mode='qp', crfs=(25,), test_mode=True))

dist_params = dict(backend='nccl')
