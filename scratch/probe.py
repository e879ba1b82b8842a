import torch, ctypes, os, time
t0=time.time()
lib=ctypes.CDLL(os.path.join(os.path.dirname(__file__),"libprobe.so"))
print("dev", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
n=1<<20
x=torch.randn(n,device="cuda"); y=torch.randn(n,device="cuda"); ref=2.5*x+y
s=torch.cuda.current_stream().cuda_stream
r=lib.probe_axpy(ctypes.c_void_p(x.data_ptr()),ctypes.c_void_p(y.data_ptr()),ctypes.c_float(2.5),ctypes.c_int(n),ctypes.c_void_p(s))
torch.cuda.synchronize(); print("axpy rc",r,"maxerr",(y-ref).abs().max().item())
A=torch.randn(16,32,device="cuda").bfloat16(); Bt=torch.randn(16,32,device="cuda").bfloat16(); C=torch.zeros(16,16,device="cuda")
r=lib.probe_mfma(ctypes.c_void_p(A.data_ptr()),ctypes.c_void_p(Bt.data_ptr()),ctypes.c_void_p(C.data_ptr()),ctypes.c_void_p(s)); torch.cuda.synchronize()
print("mfma rc",r,"maxerr",(C-A.float()@Bt.float().T).abs().max().item())
A=torch.randn(32,2,device="cuda"); B=torch.randn(2,32,device="cuda"); C=torch.zeros(32,32,device="cuda")
r=lib.probe_mfma32(ctypes.c_void_p(A.data_ptr()),ctypes.c_void_p(B.data_ptr()),ctypes.c_void_p(C.data_ptr()),ctypes.c_void_p(s)); torch.cuda.synchronize()
print("mfma32 rc",r,"maxerr",(C-A@B).abs().max().item())
# side stream
st=torch.cuda.Stream()
with torch.cuda.stream(st):
    y2=torch.zeros(n,device="cuda")
    r=lib.probe_axpy(ctypes.c_void_p(x.data_ptr()),ctypes.c_void_p(y2.data_ptr()),ctypes.c_float(1.0),ctypes.c_int(n),ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
st.synchronize(); print("side stream maxerr",(y2-x).abs().max().item())
import subprocess; print(subprocess.run("nproc; lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket'; free -g | head -2; rocminfo | grep -E 'Marketing|Compute Unit|Max Clock' | head -8",shell=True,capture_output=True,text=True).stdout)
print("elapsed",time.time()-t0)
