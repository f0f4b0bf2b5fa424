# Reference: configs/HR_davis_LR_128x128_IPB_LR_test.py (inherits the IPB config, vsr=False):
# enhancement of the 180x320 LR frames at mixed CRFs.
_base_ = ['HR_davis_LR_128x128_IPB.py']
exp_name = 'HR_davis_LR_128x128_IPB_LR_test'
data = dict(test=dict(height=180, width=320, crfs=(15, 25, 35)))
