# Reference: configs/HR_davis_LR_128x128_IPB.py -- same generator; the data layer feeds
# ord(slice)/255 as "QPs" (replace_qp_withIPB=True, :124) which is qp_mode='ipb' here.
_base_ = ['HR_davis_LR_128x128.py']
exp_name = 'HR_davis_LR_128x128_IPB'
data = dict(test=dict(qp_mode='ipb', height=720, width=1280, crfs=(25,)))
