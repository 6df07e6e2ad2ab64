// stand-in: OpenCV is absent; the compiled reference sources only include this header for debug display code that is never reached
#pragma once
