"""Drop-in counterparts of the reference's pytorch_models package: same class
names, constructor (``Model(hyper_params)``), ``forward(data)`` 7-slot contract
and ``state_dict()`` keys; every arithmetic op runs in libr4r_hip.so."""
