import sys, os
sys.path.insert(0, "/root/repo")
import torch
from robosat_amd import ops, _lib
DEV="cuda:0"
x = torch.randn(2,16,16,64, device=DEV); wt = torch.randn(256,1,1,64, device=DEV)*0.05
print("plain", ops.conv2d(x, wt).shape)
s = torch.rand(256, device=DEV)+0.5; b = torch.randn(256, device=DEV); r = torch.randn(2,16,16,256, device=DEV)
print("knobs", ops.get_knob("conv1x1_np"), ops.get_knob("conv1x1_ew"))
print("full", ops.conv2d(x, wt, scale=s, shift=b, residual=r, relu=True).shape)
with ops.knob("conv1x1_ew", 0):
    print("ew0", ops.conv2d(x, wt, scale=s, shift=b, residual=r, relu=True).shape)
with ops.knob("conv1x1_np", 0):
    print("np0", ops.conv2d(x, wt, scale=s, shift=b, residual=r, relu=True).shape)
with ops.knob("conv1x1_np", 1), ops.knob("conv1x1_ew", 0):
    print("np1", ops.conv2d(x, wt, scale=s, shift=b, residual=r, relu=True).shape)
