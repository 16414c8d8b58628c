import sys, time, numpy as np
sys.path.insert(0, '.')
import gnn_amd
dims=[784,300,100,10]; B=128; nb=64
rng=np.random.default_rng(3)
W=rng.standard_normal((784,10))
X=rng.random((B*nb,784))*(rng.random((B*nb,784))<0.19)
lab=(X@W).argmax(1); Y=np.eye(10)[lab]
net=gnn_amd.SoftmaxCrossEntropyNeuralNet(dims,max_batch=B)
net.set_weights(net.get_weights()*0.2)
net.upload_dataset(X,Y)
l0=net.loss_range(0,B).mean()
t0=time.perf_counter(); net.train_range(0,B,500000,0.0125,0.9); net.synchronize(); dt=time.perf_counter()-t0
l1=net.loss_range(0,B).mean(); w=net.get_weights()
acc=(net.argmax_range(0,B)==lab[:B]).mean()
print("500k steps in %.2f s = %.2f us/step; loss %.4f -> %.6f; weights finite: %s; max|w| %.3f; train acc on batch 0: %.3f; time=%d"%(dt,dt/5e5*1e6,l0,l1,np.isfinite(w).all(),np.abs(w).max(),acc,net.time))
