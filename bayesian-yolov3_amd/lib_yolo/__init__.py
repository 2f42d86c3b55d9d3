"""Host-side mirror of the reference's ``lib_yolo`` package for the inference path: same module,
class and function names, argument meaning and error behaviour (`lib_yolo/{yolov3,model,darknet,
data}.py` of flkraus/bayesian-yolov3), with the TensorFlow graph replaced by the graph builder of
libbyolo.so (include/byolo.h).  Training-only parts of the reference are out of scope."""
