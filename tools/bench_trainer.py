import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_amd
from gnn_amd import trainer as tr
dims=[784,300,100,10]; n=6000
rng=np.random.default_rng(0)
X=rng.random((n,784))*(rng.random((n,784))<0.19); Y=np.eye(10)[rng.integers(0,10,n)]
net=gnn_amd.SoftmaxCrossEntropyNeuralNet(dims,max_batch=128)
t=tr.NeuralNetTrainer(X, Y, net)
t.train(200, 0.0125, 128, 0.9); net.synchronize()
t0=time.perf_counter(); t.train(3000, 0.0125, 128, 0.9); net.synchronize(); dt=time.perf_counter()-t0
print("train_sampled: %.2f us/step (%.3g samples/s)"%(dt/3000*1e6, 3000*128/dt))
