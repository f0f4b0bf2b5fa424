# On-disk evaluation with the reference's directory layout and dataset type name
# (configs/HR_davis_LR_128x128.py:109-131,198-206 of ZeldaM1/PnP-VCVE); point --testdir_lr / --testdir_gt
# (or data.test.lq_folder / gt_folder) at real folders.  MV records are rasterised on the GPU.
_base_ = ['HR_davis_LR_128x128.py']
exp_name = 'REDS_folder_example'
test_pipeline = [
    dict(type='GenerateSegmentIndices_LR', interval_list=[1]),
    dict(type='LoadImageFromFileList_ipb', io_backend='disk', key='lq', channel_order='rgb', random_compress=False,
         load_mv=True, load_qp_slice=True, load_base_qp=True, load_partition=True, drconv=True,
         qp_slice_file='dataset/REDS_test_HR/multi_cprs_REDS_test_HR.json'),
    dict(type='LoadImageFromFileList', io_backend='disk', key='gt', channel_order='rgb'),
    dict(type='RescaleToZeroOne', keys=['lq', 'gt', 'base_QPs', 'QPs', 'partitions']),
    dict(type='FramesToTensor', keys=['lq', 'gt', 'mvs', 'slices', 'base_QPs', 'QPs', 'partitions']),
    dict(type='Collect', keys=['lq', 'gt', 'mvs', 'slices', 'base_QPs', 'QPs', 'partitions'],
         meta_keys=['lq_path', 'gt_path', 'key']),
]
data = dict(test=dict(_delete_=True, type='SRREDSMultipleGTCompressDataset',
                      lq_folder='dataset/REDS_test_HR/crf35/png', gt_folder='dataset/REDS_test_HR/X4/png',
                      num_input_frames=100, pipeline=test_pipeline, scale=1, val_partition='REDS4', test_mode=True))
