"""Adam written from the paper (Kingma & Ba 2015, Algorithm 1 and the 'epsilon-hat' remark at the end of section 2), independently of
oracle/avsr_oracle.py: test infrastructure, used only to cross-check the oracle's optimiser restatement.

Algorithm 1:   m_t = b1 m_{t-1} + (1-b1) g_t ;  v_t = b2 v_{t-1} + (1-b2) g_t^2
               mhat = m_t / (1-b1^t) ;  vhat = v_t / (1-b2^t) ;  theta_t = theta_{t-1} - alpha * mhat / (sqrt(vhat) + eps)
Section 2, last paragraph: the efficient form  alpha_t = alpha sqrt(1-b2^t)/(1-b1^t);  theta_t = theta_{t-1} - alpha_t m_t / (sqrt(v_t) + eps_hat)
The two differ only in where epsilon sits: eps_hat = eps * sqrt(1-b2^t).  tf.train.AdamOptimizer documents the second form with its
`epsilon` argument playing eps_hat."""
import numpy as np


def adam_algorithm1(theta, grads, alpha=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """Runs len(grads) steps of Algorithm 1 in float64; returns the parameter after every step."""
    theta = np.array(theta, dtype=np.float64)
    m = np.zeros_like(theta)
    v = np.zeros_like(theta)
    out = []
    for t, g in enumerate(grads, start=1):
        g = np.asarray(g, dtype=np.float64)
        m = b1 * m + (1.0 - b1) * g
        v = b2 * v + (1.0 - b2) * g * g
        mhat = m / (1.0 - b1 ** t)
        vhat = v / (1.0 - b2 ** t)
        theta = theta - alpha * mhat / (np.sqrt(vhat) + eps)
        out.append(theta.copy())
    return out


def adam_epsilon_hat(theta, grads, alpha=1e-3, b1=0.9, b2=0.999, eps_hat=1e-8):
    """The 'efficient' ordering of section 2 with epsilon-hat (what TF implements)."""
    theta = np.array(theta, dtype=np.float64)
    m = np.zeros_like(theta)
    v = np.zeros_like(theta)
    out = []
    for t, g in enumerate(grads, start=1):
        g = np.asarray(g, dtype=np.float64)
        m = b1 * m + (1.0 - b1) * g
        v = b2 * v + (1.0 - b2) * g * g
        alpha_t = alpha * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        theta = theta - alpha_t * m / (np.sqrt(v) + eps_hat)
        out.append(theta.copy())
    return out
