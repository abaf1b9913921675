import ctypes, torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import _lib
lib=_lib.load()
print('before torch init:', lib.pzn_device_check())
print(torch.cuda.is_available())
print('after is_available:', lib.pzn_device_check())
x=torch.zeros(1,device='cuda')
print('after tensor:', lib.pzn_device_check())
print(torch.cuda.get_device_properties(0).gcnArchName)
import subprocess
print(open('/proc/self/maps').read().count('libamdhip64'))
print([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][:3])
